cd $GRAFT_REPO_ROOT
timeout 120 python tools/dbg/x3_time.py 2>&1 | grep GEMM
( timeout 900 python -m pytest tests/test_gemm_gpu.py "tests/test_fuzz_slices_gpu.py::test_fuzz_slice_gemm_shapes_and_operand_forms" -q -m gpu 2>&1 | tail -3 )
for i in 1 2; do
( timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-200 )
( T2V_F32_GEMM=native timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-200 )
done
