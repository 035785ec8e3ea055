# round 4: bf16 captured training step (Adam-maintained copies across the warm-up restore); full suite; smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16_train.py -x -q -k "hipgraph_replay or adam_writes" > gpurun_out/r4_t28.log 2>&1 || { tail -40 gpurun_out/r4_t28.log; exit 1; }
tail -2 gpurun_out/r4_t28.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_full4.log 2>&1 || { tail -60 gpurun_out/r4_full4.log; exit 1; }
tail -3 gpurun_out/r4_full4.log
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
