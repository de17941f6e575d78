export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-secondary --no-graph > /tmp/prof_e.log 2>&1
KT=$(find /tmp/prof_e -name '*kernel_trace.csv' | head -1)
python $REPO/tools/steady_profile.py "$KT" 6 5 | tail -16
