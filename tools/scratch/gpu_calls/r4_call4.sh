cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -k "large_tile or test_conv_bf16 or two_pass or attention_core_bf16 or statistics_from" > gpurun_out/r4_t4.log 2>&1 || { tail -30 gpurun_out/r4_t4.log; exit 1; }
tail -2 gpurun_out/r4_t4.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 32x128x128x128x128 8x32x32x1024x1024 -- - SDM_BF16_8PH=0 > gpurun_out/r4_tile_ab2.log 2>&1
cat gpurun_out/r4_tile_ab2.log
timeout -k 10 900 python tools/bf16_oracle_probe.py > gpurun_out/r4_bf16_oracle_probe.log 2>&1; cat gpurun_out/r4_bf16_oracle_probe.log | tail -40
