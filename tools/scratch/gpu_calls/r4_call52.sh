# round 4 (final state): full GPU suite, smoke, bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests_final2.log 2>&1; rc=$?; tail -3 gpurun_out/r4_gpu_tests_final2.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_final2.json 2> gpurun_out/r4_bench_final2.err || exit 1
grep "\[bench\]" gpurun_out/r4_bench_final2.err | cut -c1-110 | head -14
