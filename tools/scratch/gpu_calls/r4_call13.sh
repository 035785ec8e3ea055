# round 4: feed microbenchmark against the row pitch (L2 channel camping at power-of-two pitches?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ for p in 1024 1152 1280 2048 2176 9216 512 640; do timeout -k 10 120 tools/scratch/micro/feed 576 2048 $p || exit 1; done; } > gpurun_out/r4_feed_pitch.log 2>&1
grep "CUs\|LDS-DMA, 8\|register-staged + ds_write_b128, 4\|8 in flight" gpurun_out/r4_feed_pitch.log | cut -c1-150
