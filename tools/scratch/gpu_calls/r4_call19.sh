# round 4: wgrad8x3 with the three kernel rows of a slice on one XCD: parity, timing, HBM traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_train.py -x -q -k "three_tap" > gpurun_out/r4_t19.log 2>&1 || { tail -40 gpurun_out/r4_t19.log; exit 1; }
tail -2 gpurun_out/r4_t19.log
timeout -k 10 600 python tools/wgrad_ab.py 8x256x256x128x128 8x128x128x256x256 8x64x64x512x512 32x128x128x128x128 -- SDM_WGRAD8=0 - > gpurun_out/r4_wgrad_ab2.log 2>&1; cat gpurun_out/r4_wgrad_ab2.log
bash tools/profile_round4.sh pmc_wgrad > gpurun_out/r4_pmc_w2.txt 2>&1; grep -i "tflops\|hbm_read\|traffic_over\|mfma_busy\|duration_ns_mean" gpurun_out/r4_pmc_w2.txt
timeout -k 10 300 python tools/train_probe.py c4 bf16 8 2>&1 | tail -1
timeout -k 10 300 python tools/train_probe.py c5 bf16 8 2>&1 | tail -1
