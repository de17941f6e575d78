cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/dbg/x3_time.py 2>&1 | grep GEMM
( timeout 1500 python -m pytest tests/test_gemm_gpu.py "tests/test_fuzz_slices_gpu.py::test_fuzz_slice_gemm_shapes_and_operand_forms" -q -m gpu -s 2>&1 ) > gpurun_out/r06_c3_gemm_full.txt 2>&1
grep -E "^\.*F*GEMM|^dW GEMM|passed|failed|^FAILED|^E  " gpurun_out/r06_c3_gemm_full.txt
