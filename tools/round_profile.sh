# usage: bash tools/round_profile.sh <tag>   (on the GPU box) -> gpurun_out/<tag>_*.{csv,txt}
TAG=${1:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary > /tmp/prof_bench_$TAG.log 2>&1
tail -1 /tmp/prof_bench_$TAG.log | cut -c1-300
ST=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); KT=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
cp "$ST" $REPO/gpurun_out/${TAG}_bench_kernel_stats.csv
T2V_PROFILE_SEQ=$REPO/gpurun_out/${TAG}_step_sequence.txt python $REPO/tools/steady_profile.py "$KT" 6 45 $REPO/gpurun_out/${TAG}_kernel_durations.json > $REPO/gpurun_out/${TAG}_steady_state.txt
cd $REPO
# the same trace with eager launches (rocprofv3's per-kernel tracing stalls graph replays for ~1 ms every few dozen nodes,
# so the graph-mode trace overstates the idle time; GPU-busy time per kernel is unaffected)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_eager -o ${TAG}e -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary --no-graph > /tmp/prof_bench_${TAG}_eager.log 2>&1
KT=$(find /tmp/prof_${TAG}_eager -name '*kernel_trace.csv' | head -1)
python $REPO/tools/steady_profile.py "$KT" 6 45 > $REPO/gpurun_out/${TAG}_steady_state_eager.txt
cd $REPO
