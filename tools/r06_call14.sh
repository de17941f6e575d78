cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_bf16_gpu.py tests/test_decoder_persist16_gpu.py "tests/test_fuzz_slices_gpu.py::test_fuzz_slice_gemm_shapes_and_operand_forms" -q -m gpu 2>&1 | tail -5 )
for i in 1 2; do
for gr in "1 1" "0 0"; do
set -- $gr
echo "T2V_BF16_GEMM_PLANES=$1 T2V_DW_GROUPED=$2"; ( T2V_BF16_GEMM_PLANES=$1 T2V_DW_GROUPED=$2 timeout 300 python bench.py --bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-200 )
done; done
