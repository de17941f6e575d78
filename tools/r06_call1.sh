# round 6, first GPU call: the new tests + x3 GEMM accuracy / throughput + a quick bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_fuzz_slices_gpu.py -q -m gpu -s 2>&1 | tail -60 ) > gpurun_out/r06_c1_gemm.txt 2>&1
( timeout 1500 python -m pytest "tests/test_bf16_gpu.py::test_config5_shape_both_builds_against_the_cpu_oracle" "tests/test_train_plumbing_gpu.py::test_timeout_on_the_first_step_after_a_resume_restores_the_loaded_batchnorm_statistics" "tests/test_train_plumbing_gpu.py::test_replay_watchdog_drops_a_graph_that_is_slower_than_the_eager_step" "tests/test_train_plumbing_gpu.py::test_persistent_timeout_skips_the_update_and_the_engine_reruns_the_step" tests/test_ddp_two_ranks_gpu.py -q -m gpu -s 2>&1 | tail -60 ) > gpurun_out/r06_c1_newtests.txt 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-secondary 2>&1 | tail -3 | cut -c1-1500 ) > gpurun_out/r06_c1_bench_x3.txt 2>&1
( T2V_F32_GEMM=native timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-secondary 2>&1 | tail -3 | cut -c1-1500 ) > gpurun_out/r06_c1_bench_native.txt 2>&1
tail -5 gpurun_out/r06_c1_gemm.txt gpurun_out/r06_c1_newtests.txt
cut -c1-400 gpurun_out/r06_c1_bench_x3.txt gpurun_out/r06_c1_bench_native.txt
