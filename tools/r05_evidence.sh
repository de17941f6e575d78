# Round-5 evidence of the CURRENT code, on the GPU box: bash tools/r05_evidence.sh [quick]   -> gpurun_out/r05_*
# quick: tests + bench lines + traces only (no PMC passes, no stamps)
MODE=${1:-full}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r05_gpu_tests.txt 2>&1
tail -3 gpurun_out/r05_gpu_tests.txt
python bench.py 2> gpurun_out/r05_bench_err.log | tail -1 > gpurun_out/r05_bench_line.json
cut -c1-300 gpurun_out/r05_bench_line.json
python bench.py --bf16 --no-cpu-baseline --no-decode --no-secondary 2>> gpurun_out/r05_bench_err.log | tail -1 > gpurun_out/r05_bench_line_bf16.json
cut -c1-300 gpurun_out/r05_bench_line_bf16.json
bash tools/round_profile.sh r05 > gpurun_out/r05_round_profile.log 2>&1
bash tools/bf16_profile.sh > gpurun_out/r05_bf16_profile.log 2>&1
cp gpurun_out/bf16_steady_state.txt gpurun_out/r05_steady_state_bf16.txt; cp gpurun_out/bf16_step_sequence.txt gpurun_out/r05_step_sequence_bf16.txt
cp gpurun_out/bf16_kernel_durations.json gpurun_out/r05_kernel_durations_bf16.json 2>/dev/null
bash tools/koemo_profile.sh > gpurun_out/r05_koemo_profile.log 2>&1
cp gpurun_out/koemo_steady_state.txt gpurun_out/r05_steady_state_koemo.txt; cp gpurun_out/koemo_step_sequence.txt gpurun_out/r05_step_sequence_koemo.txt
head -8 gpurun_out/r05_steady_state.txt; head -6 gpurun_out/r05_steady_state_bf16.txt; head -4 gpurun_out/r05_steady_state_koemo.txt
if [ "$MODE" = "full" ]; then
  bash tools/pmc_fetch_size.sh r05 > /dev/null 2>&1
  bash tools/pmc_fetch_size.sh r05 --bf16 > /dev/null 2>&1
  bash tools/pmc_mfma.sh r05 > /dev/null 2>&1
  bash tools/pmc_mfma.sh r05 --bf16 > /dev/null 2>&1
  T2V_STAMP_ONLY=step_begin,dec_fwd_begin,dec_fwd_end,postnet_fwd_end,loss_end,dec_bwd_begin,dec_bwd_end,step_end python tools/stamps.py > gpurun_out/r05_phase_stamps.txt 2>&1
  T2V_STAMP_ONLY=step_begin,dec_bwd_end,side_vae_end,side_w_end,side_g_end,step_end python tools/stamps.py >> gpurun_out/r05_phase_stamps.txt 2>&1
  T2V_STAMP_ONLY=step_begin,dec_fwd_begin,dec_fwd_end,postnet_fwd_end,loss_end,dec_bwd_begin,dec_bwd_end,step_end python tools/stamps.py --bf16 > gpurun_out/r05_phase_stamps_bf16.txt 2>&1
  T2V_STAMP_ONLY=step_begin,dec_bwd_end,side_vae_end,side_w_end,side_g_end,step_end python tools/stamps.py --bf16 >> gpurun_out/r05_phase_stamps_bf16.txt 2>&1
  python tools/dbg/persist16_prof.py 16 84 400 > gpurun_out/r05_persist16_fwd_timeline.txt 2>&1
  python tools/dbg/persist16_bwd_prof.py 16 84 400 > gpurun_out/r05_persist16_bwd_timeline.txt 2>&1
  # the captured DAG of both steps: critical path (perfect scheduler, traced durations) and per-node ready-vs-start waits
  export TMPDIR=/tmp
  REPO=$(pwd)
  for cfg in f32 bf16; do
    FLAG=""; ANCHOR="void k_achain_bwd"; [ $cfg = bf16 ] && FLAG="--bf16" && ANCHOR="k_bwd_persist16"
    rm -rf /tmp/prof_g
    (cd /tmp && T2V_GRAPH_DOT=$REPO/gpurun_out/r05_step_graph_$cfg.dot timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o g -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary $FLAG > /tmp/prof_g.log 2>&1)
    KT=$(find /tmp/prof_g -name '*kernel_trace.csv' | head -1)
    python tools/graph_critical_path.py gpurun_out/r05_step_graph_$cfg.dot $KT "$ANCHOR" > gpurun_out/r05_critical_path_$cfg.txt 2>&1
    python tools/graph_node_waits.py gpurun_out/r05_step_graph_$cfg.dot $KT 40 > gpurun_out/r05_node_waits_$cfg.txt 2>&1
  done
  T2V_STAMP_ONLY=step_begin,dec_bwd_end,bilstm_bwd_begin,bilstm_bwd_end,bwd_main_end,grads_ready,step_end python tools/stamps.py >> gpurun_out/r05_phase_stamps.txt 2>&1
  T2V_STAMP_ONLY=step_begin,dec_bwd_end,bilstm_bwd_begin,bilstm_bwd_end,bwd_main_end,grads_ready,step_end python tools/stamps.py --bf16 >> gpurun_out/r05_phase_stamps_bf16.txt 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && timeout 120 /tmp/mfma_peak > gpurun_out/r05_mfma_peak.txt 2>&1
  python tools/dbg/gemm_time.py > gpurun_out/r05_gemm_time.txt 2>&1
  python tools/dbg/gemm_time.py --bf16 >> gpurun_out/r05_gemm_time.txt 2>&1
  grep -h "graph step\|kernel k_\|critical path [0-9]" gpurun_out/r05_phase_stamps*.txt gpurun_out/r05_persist16_*_timeline.txt gpurun_out/r05_critical_path_*.txt
fi
