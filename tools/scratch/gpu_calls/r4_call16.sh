# round 4: feed microbenchmark, every workgroup's k walk rotated (de-correlates the 128-B column all CUs ask for at one time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ for w in 64 256 512; do for r in 0 1; do timeout -k 10 120 tools/scratch/micro/feed 576 $w 1024 128 $r || exit 1; done; done; } > gpurun_out/r4_feed_rot.log 2>&1
grep "CUs\|LDS-DMA, 8\|8 waves, 8 in flight" gpurun_out/r4_feed_rot.log | cut -c1-170
