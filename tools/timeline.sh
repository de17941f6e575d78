# usage (GPU box): bash tools/timeline.sh <tag> [bench args]  -> gpurun_out/<tag>_timeline.txt
TAG=${1:-tl}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-secondary "$@" > /tmp/prof_bench_$TAG.log 2>&1
tail -1 /tmp/prof_bench_$TAG.log | cut -c1-200
KT=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python $REPO/tools/timeline.py "$KT" $REPO/gpurun_out/${TAG}_timeline.txt
tail -1 $REPO/gpurun_out/${TAG}_timeline.txt
cd $REPO
