cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_fuzz_slices_gpu.py -q -m gpu -s 2>&1 ) > gpurun_out/r06_c2_gemm_full.txt 2>&1
grep -E "^GEMM|^dW GEMM|passed|failed|^FAILED|^E  " gpurun_out/r06_c2_gemm_full.txt > gpurun_out/r06_c2_gemm.txt
cat gpurun_out/r06_c2_gemm.txt
