export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary --bf16 > /tmp/prof_b.log 2>&1
tail -1 /tmp/prof_b.log | cut -c1-200
KT=$(find /tmp/prof_b -name '*kernel_trace.csv' | head -1)
T2V_PROFILE_SEQ=$REPO/gpurun_out/bf16_step_sequence.txt python $REPO/tools/steady_profile.py "$KT" 6 28 $REPO/gpurun_out/bf16_kernel_durations.json > $REPO/gpurun_out/bf16_steady_state.txt
