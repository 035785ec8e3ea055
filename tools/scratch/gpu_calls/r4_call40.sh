# round 4: 8-wave 256x128 row-patch tile for the 128-channel 3x3 convs: parity, then A/B on the conv and the C3 forward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "test_conv_bf16 or epilogue or row_patch" > gpurun_out/r4_row256_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r4_row256_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
L=gpurun_out/r04_row256_ab.txt; : > $L
timeout -k 10 600 python tools/tile_ab.py 128x128x128x128x128 32x256x256x128x128 32x128x128x128x128 64x64x64x128x128 -- SDM_BF16_ROW256=0 - SDM_BF16_ROW256=0 - 2>/dev/null >> $L || exit 1
for v in 1 0 1 0; do
  echo "# SDM_BF16_ROW256=$v" >> $L
  SDM_BF16_ROW256=$v timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 10 2>/dev/null | grep forward >> $L || exit 1
done
cat $L
