set -x
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python /root/repo/bench.py --steps 8 --warmup 3 > /tmp/prof_bench.log 2>&1
tail -1 /tmp/prof_bench.log | cut -c1-200
ST=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); KT=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
cp "$ST" /root/repo/gpurun_out/r01_bench_kernel_stats.csv
python /root/repo/tools/steady_profile.py "$KT" 6 40 > /root/repo/gpurun_out/r01_steady_state.txt
cd /root/repo && timeout 300 python bench.py > gpurun_out/r01_bench_line.json 2>gpurun_out/bench_err.log; tail -c 600 gpurun_out/r01_bench_line.json
