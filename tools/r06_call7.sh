cd $GRAFT_REPO_ROOT
timeout 120 python tools/dbg/conv_x3_time.py 2>&1 | grep conv
for sp in 1 2 4 8; do echo "T2V_CONV_X3_SPLITS=$sp"; T2V_CONV_X3_SPLITS=$sp timeout 120 python tools/dbg/conv_x3_time.py 2>&1 | grep conv | cut -c1-60; done
timeout 120 python tools/dbg/x3_time.py 2>&1 | grep GEMM
( timeout 900 python -m pytest tests/test_conv_bn_gpu.py tests/test_gemm_gpu.py -q -m gpu -s 2>&1 ) > gpurun_out/r06_c7_full.txt 2>&1
grep -E "passed|failed|^FAILED|^E  " gpurun_out/r06_c7_full.txt
