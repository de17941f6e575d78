cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1
tail -2 gpurun_out/r06_gpu_tests.txt
timeout 600 python bench.py 2> gpurun_out/r06_bench_err.log | tail -1 > gpurun_out/r06_bench_line.json
cut -c1-400 gpurun_out/r06_bench_line.json
timeout 600 python bench.py --bf16 --no-cpu-baseline --no-decode --no-secondary 2>> gpurun_out/r06_bench_err.log | tail -1 > gpurun_out/r06_bench_line_bf16.json
cut -c1-300 gpurun_out/r06_bench_line_bf16.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
