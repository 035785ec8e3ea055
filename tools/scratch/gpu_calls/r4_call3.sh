cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "sampler_requires" > gpurun_out/r4_t3.log 2>&1 || { tail -20 gpurun_out/r4_t3.log; exit 1; }
tail -2 gpurun_out/r4_t3.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 32x128x128x128x128 8x32x32x1024x1024 8x256x256x128x128 -- - SDM_BF16_8PH=0 > gpurun_out/r4_tile_ab.log 2>&1
cat gpurun_out/r4_tile_ab.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err; tail -c 6000 gpurun_out/r4_bench1.json
