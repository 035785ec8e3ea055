# round 4: igemm8 with the taps innermost in the K walk: parity (forced tiles, goldens), A/B timing, counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_train.py -x -q -k 'conv_fwd_dgrad_wgrad or concat or token_linear or attention_core_backward or grads_golden or three_tap or wgrad' > gpurun_out/r4_t20a.log 2>&1 || { tail -40 gpurun_out/r4_t20a.log; exit 1; }
tail -2 gpurun_out/r4_t20a.log
SDM_BF16_TAPIN=1 SDM_BF16_FORCE=2,1 timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py -x -q -k "test_conv_bf16 or forward_bf16_vs_golden or residual_block or statistics_from or fwd_dgrad_wgrad or concat or two_pass or attention_core" > gpurun_out/r4_t20.log 2>&1 || { tail -40 gpurun_out/r4_t20.log; exit 1; }
tail -2 gpurun_out/r4_t20.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 8x32x32x1024x1024 8x64x64x1024x512 -- - SDM_BF16_TAPIN=1 > gpurun_out/r4_tile_tapin.log 2>&1; cat gpurun_out/r4_tile_tapin.log
bash tools/profile_round4.sh pmc_conv_tapin > gpurun_out/r4_pmc_tapin.txt 2>&1; grep -i "tflops\|hbm_read\|traffic_over\|mfma_busy\|duration_ns_mean\|waitcnt\|valu_per" gpurun_out/r4_pmc_tapin.txt
for v in 0 1; do echo "== SDM_BF16_TAPIN=$v"; SDM_BF16_TAPIN=$v timeout -k 10 300 python tools/train_probe.py c4 bf16 8 2>&1 | tail -1; SDM_BF16_TAPIN=$v timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 4 2>&1 | tail -1; done
timeout -k 10 300 python tools/train_probe.py c2 f32 30 2>&1 | tail -1
timeout -k 10 600 python tools/wgrad_ab.py 8x128x128x256x256 8x64x64x512x512 -- SDM_WGRAD3=0,SDM_WGRAD8=0 > gpurun_out/r4_wgrad_ab3.log 2>&1; cat gpurun_out/r4_wgrad_ab3.log
