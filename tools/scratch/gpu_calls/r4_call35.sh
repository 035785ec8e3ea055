# round 4: full GPU suite + bench line after the edge kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests_edge.log 2>&1; rc=$?; tail -4 gpurun_out/r4_gpu_tests_edge.log | cut -c1-300
[ $rc -eq 0 ] && timeout -k 10 900 python bench.py > gpurun_out/r4_bench_edge.json 2> gpurun_out/r4_bench_edge.err; tail -c 3000 gpurun_out/r4_bench_edge.json
