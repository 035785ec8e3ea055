# round 4: full GPU suite; k-walk rotation A/B + parity under it; headline profile set (kernel stats, HBM traffic, launch dump); sampler / C4 kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k 'concurrent_forwards or embedding or unet_grads_golden or linear' > gpurun_out/r4_full2.log 2>&1 || { tail -60 gpurun_out/r4_full2.log; exit 1; }
tail -3 gpurun_out/r4_full2.log
SDM_BF16_KROT=1 SDM_BF16_FORCE=2,1 timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "test_conv_bf16 or forward_bf16_vs_golden or residual_block or statistics_from" > gpurun_out/r4_t17.log 2>&1 || { tail -40 gpurun_out/r4_t17.log; exit 1; }
tail -2 gpurun_out/r4_t17.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 8x32x32x1024x1024 8x64x64x1024x512 -- - SDM_BF16_KROT=1 > gpurun_out/r4_tile_krot.log 2>&1; cat gpurun_out/r4_tile_krot.log
timeout -k 10 300 python tools/mall_probe.py > gpurun_out/r4_mall_probe.log 2>&1; timeout -k 10 300 python tools/mall_probe.py 1 16 16 512 512 >> gpurun_out/r4_mall_probe.log 2>&1; cat gpurun_out/r4_mall_probe.log
LINES_SHOWN=12 bash tools/profile_round4.sh train traffic dump sample c4 > gpurun_out/r4_prof_c.txt 2>&1; tail -80 gpurun_out/r4_prof_c.txt | cut -c1-170
