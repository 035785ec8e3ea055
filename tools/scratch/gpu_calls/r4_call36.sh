# round 4: stem kernel up to 8 input channels + its data-gradient mode (head dgrad): tests, C4/C5 bf16 training time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r4_edge_tests2.log 2>&1; rc=$?; tail -5 gpurun_out/r4_edge_tests2.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r4_bench_edge2.json 2> gpurun_out/r4_bench_edge2.err; tail -c 1500 gpurun_out/r4_bench_edge2.json; tail -3 gpurun_out/r4_bench_edge2.err
