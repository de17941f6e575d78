cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
REPO=$(pwd)
for cfg in f32; do
  FLAG=""; ANCHOR="void k_achain_bwd"; [ $cfg = bf16 ] && FLAG="--bf16" && ANCHOR="k_bwd_persist16"
  rm -rf /tmp/prof_g
  (cd /tmp && T2V_GRAPH_DOT=$REPO/gpurun_out/r06_step_graph_$cfg.dot timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o g -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary $FLAG > /tmp/prof_g.log 2>&1)
  KT=$(find /tmp/prof_g -name '*kernel_trace.csv' | head -1)
  python tools/graph_critical_path.py gpurun_out/r06_step_graph_$cfg.dot $KT "$ANCHOR" > gpurun_out/r06_critical_path_$cfg.txt 2>&1
  python tools/graph_node_waits.py gpurun_out/r06_step_graph_$cfg.dot $KT 40 > gpurun_out/r06_node_waits_$cfg.txt 2>&1
done
tail -60 gpurun_out/r06_critical_path_f32.txt
