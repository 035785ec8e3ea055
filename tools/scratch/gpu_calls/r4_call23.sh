# round 4: taps-innermost igemm8 with the address generation under the MFMAs: parity, A/B; then the gate-repeat flake check
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SDM_BF16_TAPIN=1 SDM_BF16_FORCE=2,1 timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py -x -q -k "test_conv_bf16 or forward_bf16_vs_golden or residual_block or statistics_from or fwd_dgrad_wgrad or concat or two_pass or attention_core" > gpurun_out/r4_t23.log 2>&1 || { tail -40 gpurun_out/r4_t23.log; exit 1; }
tail -2 gpurun_out/r4_t23.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 8x32x32x1024x1024 8x64x64x1024x512 -- - SDM_BF16_TAPIN=1 > gpurun_out/r4_tile_tapin2.log 2>&1; cat gpurun_out/r4_tile_tapin2.log
for v in 0 1; do echo "== SDM_BF16_TAPIN=$v"; SDM_BF16_TAPIN=$v timeout -k 10 300 python tools/train_probe.py c4 bf16 8 2>&1 | tail -1; SDM_BF16_TAPIN=$v timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 4 2>&1 | tail -1; done
bash tools/scratch/gpu_calls/r4_call22.sh
