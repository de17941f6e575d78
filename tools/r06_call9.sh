cd $GRAFT_REPO_ROOT
for i in 1 2; do
for cx in 1 0; do
echo "T2V_CONV_X3=$cx"; ( T2V_CONV_X3=$cx timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-200 )
done; done
T2V_CONV_X3=0 T2V_STAMP_ONLY=step_begin,dec_fwd_begin,dec_fwd_end,postnet_fwd_end,loss_end,dec_bwd_begin,dec_bwd_end,step_end timeout 300 python tools/stamps.py 2>&1 | tail -10
T2V_CONV_X3=0 T2V_STAMP_ONLY=step_begin,dec_bwd_end,side_vae_end,side_w_end,side_g_end,step_end timeout 300 python tools/stamps.py 2>&1 | tail -7
T2V_CONV_X3=0 T2V_STAMP_ONLY=step_begin,dec_bwd_end,bilstm_bwd_begin,bilstm_bwd_end,bwd_main_end,grads_ready,step_end timeout 300 python tools/stamps.py 2>&1 | tail -8
T2V_CONV_X3=1 T2V_STAMP_ONLY=step_begin,dec_fwd_begin,dec_fwd_end,postnet_fwd_end,loss_end,dec_bwd_begin,dec_bwd_end,step_end timeout 300 python tools/stamps.py 2>&1 | tail -10
