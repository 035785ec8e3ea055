# round 4 (late): full GPU suite, smoke, bench line, refreshed kernel totals of the C3 forward and the C4 / C5 bf16 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests_final.log 2>&1; rc=$?; tail -3 gpurun_out/r4_gpu_tests_final.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err || exit 1
grep "\[bench\]" gpurun_out/r4_bench_final.err | cut -c1-110
LINES_SHOWN=6 timeout -k 10 900 bash tools/profile_round4.sh c3 c4 c5 > gpurun_out/r4_prof_final.log 2>&1; tail -24 gpurun_out/r4_prof_final.log | cut -c1-200
