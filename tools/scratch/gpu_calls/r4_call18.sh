# round 4: y_scale weight-gradient kernel with batched loads: parity + C4 / headline timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k 'embedding or unet_grads_golden or linear or standard_architecture_training' > gpurun_out/r4_t18.log 2>&1 || { tail -40 gpurun_out/r4_t18.log; exit 1; }
tail -2 gpurun_out/r4_t18.log
timeout -k 10 300 python tools/train_probe.py c4 bf16 8 2>&1 | tail -1
timeout -k 10 300 python tools/train_probe.py c5 bf16 8 2>&1 | tail -1
timeout -k 10 300 python tools/train_probe.py c2 f32 30 2>&1 | tail -1
LINES_SHOWN=16 bash tools/profile_round4.sh c4 2>&1 | grep -i "linear_small_bwd_dw\|igemm8\|wgrad" | cut -c1-150
