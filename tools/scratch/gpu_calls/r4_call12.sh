# round 4: feed-path microbenchmark; opt-in in-kernel split-K test; split parity tests with the switch off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ timeout -k 10 120 tools/scratch/micro/feed 576 512 && timeout -k 10 120 tools/scratch/micro/feed 576 4096 && timeout -k 10 120 tools/scratch/micro/feed 576 64; } > gpurun_out/r4_feed.log 2>&1
cat gpurun_out/r4_feed.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "split_k" > gpurun_out/r4_t12.log 2>&1 || { tail -40 gpurun_out/r4_t12.log; exit 1; }
tail -2 gpurun_out/r4_t12.log
