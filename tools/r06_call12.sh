cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_decoder_gpu.py tests/test_decoder_persist_train_gpu.py "tests/test_model_gpu.py" -q -m gpu 2>&1 | tail -5 )
for i in 1 2; do
for gr in 1 0; do
echo "T2V_DW_GROUPED=$gr"; ( T2V_DW_GROUPED=$gr timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-200 )
done; done
T2V_STAMP_ONLY=step_begin,dec_bwd_end,side_vae_end,side_w_end,side_g_end,step_end timeout 300 python tools/stamps.py 2>&1 | tail -7
