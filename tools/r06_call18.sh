cd $GRAFT_REPO_ROOT
T2V_STAMP_ONLY=step_begin,dec_fwd_begin,dec_fwd_end,postnet_fwd_end,loss_end,dec_bwd_begin,dec_bwd_end,step_end timeout 300 python tools/stamps.py --bf16 2>&1 | tail -9
T2V_STAMP_ONLY=step_begin,dec_bwd_end,side_vae_end,side_w_end,side_g_end,step_end timeout 300 python tools/stamps.py --bf16 2>&1 | tail -7
T2V_STAMP_ONLY=step_begin,dec_bwd_end,bilstm_bwd_begin,bilstm_bwd_end,bwd_main_end,grads_ready,step_end timeout 300 python tools/stamps.py --bf16 2>&1 | tail -8
bash tools/bf16_profile.sh > gpurun_out/r06_bf16_profile.log 2>&1; head -45 gpurun_out/bf16_steady_state.txt
