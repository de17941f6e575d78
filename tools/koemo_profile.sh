export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary --koemo > /tmp/prof_k.log 2>&1
tail -1 /tmp/prof_k.log | cut -c1-200
KT=$(find /tmp/prof_k -name '*kernel_trace.csv' | head -1)
T2V_PROFILE_SEQ=$REPO/gpurun_out/koemo_step_sequence.txt python $REPO/tools/steady_profile.py "$KT" 6 30 > $REPO/gpurun_out/koemo_steady_state.txt
