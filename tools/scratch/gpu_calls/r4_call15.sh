# round 4: is the 128-B-piece feed rate a power-of-two-pitch effect?  product kernel at 512 / 576 / 640 / 768 channels; microbench at pitches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x64x64x576x576 8x64x64x640x640 8x64x64x768x768 8x64x64x1024x1024 -- - > gpurun_out/r4_tile_pitch.log 2>&1; cat gpurun_out/r4_tile_pitch.log
{ for p in 1024 1152 1280 1536; do timeout -k 10 120 tools/scratch/micro/feed 576 96 $p 128 || exit 1; done; } > gpurun_out/r4_feed_pitch2.log 2>&1
grep "CUs\|LDS-DMA, 8" gpurun_out/r4_feed_pitch2.log | cut -c1-160
