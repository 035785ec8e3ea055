# round 4: feed microbenchmark against the contiguous bytes per row (access density), L2-resident and streaming windows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ for w in 64 4096; do for pc in 128 256 512 1024; do timeout -k 10 120 tools/scratch/micro/feed 576 $w 1024 $pc || exit 1; done; done; } > gpurun_out/r4_feed_piece.log 2>&1
grep "CUs\|LDS-DMA, 8\|register-staged + ds_write_b128, 4\|8 waves, 8 in flight" gpurun_out/r4_feed_piece.log | cut -c1-160
