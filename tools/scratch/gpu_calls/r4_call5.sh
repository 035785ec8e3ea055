# round 4 (re-entry): full GPU suite, then tile A/B of the phase-interleaved kernel, then the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_full1.log 2>&1 || { tail -40 gpurun_out/r4_full1.log; exit 1; }
tail -3 gpurun_out/r4_full1.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 32x128x128x128x128 8x32x32x1024x1024 8x256x256x128x128 -- - SDM_BF16_8PH=0 > gpurun_out/r4_tile_ab.log 2>&1 || { tail -20 gpurun_out/r4_tile_ab.log; exit 1; }
cat gpurun_out/r4_tile_ab.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err || { tail -20 gpurun_out/r4_bench1.err; exit 1; }
tail -c 3000 gpurun_out/r4_bench1.json
