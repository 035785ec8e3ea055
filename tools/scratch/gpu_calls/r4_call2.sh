cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py -x -q -k "large_tile or cooperative or concurrent_forwards or sampler_requires or test_conv_bf16 or two_pass or attention_core_bf16 or launch_plan or sampler_traj" > gpurun_out/r4_t2.log 2>&1
rc=$?
tail -25 gpurun_out/r4_t2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 32x128x128x128x128 8x32x32x1024x1024 8x256x256x128x128 -- - SDM_BF16_8PH=0 > gpurun_out/r4_tile_ab.log 2>&1
cat gpurun_out/r4_tile_ab.log
