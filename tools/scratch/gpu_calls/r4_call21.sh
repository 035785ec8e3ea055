# round 4: full GPU suite on the current tree, then the full bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_full3.log 2>&1 || { tail -60 gpurun_out/r4_full3.log; exit 1; }
tail -3 gpurun_out/r4_full3.log
timeout -k 10 1000 python bench.py > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err || { tail -20 gpurun_out/r4_bench2.err; exit 1; }
grep "bench\]" gpurun_out/r4_bench2.err | cut -c1-200
