# round 4: igemm8's fixed cost per tile: the same 256-tile launch (8 x 64x64 pixels, 512 output channels) at 64..1024 input channels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/tile_ab.py 8x64x64x64x512 8x64x64x128x512 8x64x64x256x512 8x64x64x512x512 8x64x64x1024x512 8x64x64x2048x512 -- - > gpurun_out/r4_tile_k.log 2>&1; cat gpurun_out/r4_tile_k.log
timeout -k 10 120 tools/scratch/micro/gemm8ph 32768 512 64 9 64 && timeout -k 10 120 tools/scratch/micro/gemm8ph 32768 512 128 9 64 && timeout -k 10 120 tools/scratch/micro/gemm8ph 32768 512 256 9 64 && timeout -k 10 120 tools/scratch/micro/gemm8ph 32768 512 1024 9 64
