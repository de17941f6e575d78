# Round-6 evidence of the CURRENT code, on the GPU box: bash tools/r06_evidence.sh [quick]   -> gpurun_out/r06_*
MODE=${1:-full}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1
tail -3 gpurun_out/r06_gpu_tests.txt
timeout 600 python bench.py 2> gpurun_out/r06_bench_err.log | tail -1 > gpurun_out/r06_bench_line.json
cut -c1-300 gpurun_out/r06_bench_line.json
timeout 600 python bench.py --bf16 --no-cpu-baseline --no-decode --no-secondary 2>> gpurun_out/r06_bench_err.log | tail -1 > gpurun_out/r06_bench_line_bf16.json
cut -c1-300 gpurun_out/r06_bench_line_bf16.json
bash tools/round_profile.sh r06 > gpurun_out/r06_round_profile.log 2>&1
bash tools/bf16_profile.sh > gpurun_out/r06_bf16_profile.log 2>&1
cp gpurun_out/bf16_steady_state.txt gpurun_out/r06_steady_state_bf16.txt; cp gpurun_out/bf16_step_sequence.txt gpurun_out/r06_step_sequence_bf16.txt
cp gpurun_out/bf16_kernel_durations.json gpurun_out/r06_kernel_durations_bf16.json 2>/dev/null
head -8 gpurun_out/r06_steady_state.txt; head -6 gpurun_out/r06_steady_state_bf16.txt
if [ "$MODE" = "full" ]; then
  bash tools/pmc_fetch_size.sh r06 > /dev/null 2>&1
  bash tools/pmc_fetch_size.sh r06 --bf16 > /dev/null 2>&1
  bash tools/pmc_mfma.sh r06 > /dev/null 2>&1
  bash tools/pmc_mfma.sh r06 --bf16 > /dev/null 2>&1
  for suf in "" "_bf16"; do
    FLAG=""; [ "$suf" = "_bf16" ] && FLAG="--bf16"
    T2V_STAMP_ONLY=step_begin,dec_fwd_begin,dec_fwd_end,postnet_fwd_end,loss_end,dec_bwd_begin,dec_bwd_end,step_end timeout 300 python tools/stamps.py $FLAG > gpurun_out/r06_phase_stamps$suf.txt 2>&1
    T2V_STAMP_ONLY=step_begin,dec_bwd_end,side_vae_end,side_w_end,side_g_end,step_end timeout 300 python tools/stamps.py $FLAG >> gpurun_out/r06_phase_stamps$suf.txt 2>&1
    T2V_STAMP_ONLY=step_begin,dec_bwd_end,bilstm_bwd_begin,bilstm_bwd_end,bwd_main_end,grads_ready,step_end timeout 300 python tools/stamps.py $FLAG >> gpurun_out/r06_phase_stamps$suf.txt 2>&1
  done
  # time lines of all four persistent kernels (one step in the middle of the pass, per-role medians)
  timeout 300 python tools/dbg/persist_prof.py 6 84 400 > gpurun_out/r06_fwd_persist_timeline.txt 2>&1
  timeout 300 python tools/dbg/persist_bwd_prof.py 6 84 400 > gpurun_out/r06_bwd_persist_timeline.txt 2>&1
  timeout 300 python tools/dbg/persist16_prof.py 16 84 400 > gpurun_out/r06_persist16_fwd_timeline.txt 2>&1
  timeout 300 python tools/dbg/persist16_bwd_prof.py 16 84 400 > gpurun_out/r06_persist16_bwd_timeline.txt 2>&1
  # the dense products alone
  ( timeout 200 python tools/dbg/x3_time.py; timeout 200 python tools/dbg/bf16_gemm_time.py; T2V_BF16_GEMM_PLANES=0 timeout 200 python tools/dbg/bf16_gemm_time.py;
    timeout 200 python tools/dbg/conv_x3_time.py; timeout 200 python tools/dbg/conv_bf16_time.py ) 2>&1 | grep -E "GEMM|conv" > gpurun_out/r06_gemm_time.txt
  for cfg in f32 bf16; do
    FLAG=""; ANCHOR="void k_achain_bwd"; [ $cfg = bf16 ] && FLAG="--bf16" && ANCHOR="k_bwd_persist16"
    rm -rf /tmp/prof_g
    (cd /tmp && T2V_GRAPH_DOT=$REPO/gpurun_out/r06_step_graph_$cfg.dot timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o g -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary $FLAG > /tmp/prof_g.log 2>&1)
    KT=$(find /tmp/prof_g -name '*kernel_trace.csv' | head -1)
    python tools/graph_critical_path.py gpurun_out/r06_step_graph_$cfg.dot $KT "$ANCHOR" > gpurun_out/r06_critical_path_$cfg.txt 2>&1
    python tools/graph_node_waits.py gpurun_out/r06_step_graph_$cfg.dot $KT 40 > gpurun_out/r06_node_waits_$cfg.txt 2>&1
  done
  grep -h "graph step\|critical path [0-9]" gpurun_out/r06_phase_stamps*.txt gpurun_out/r06_critical_path_*.txt
fi
