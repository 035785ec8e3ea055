# round 4: persistent workgroups in the row-patch kernel: parity (forced row tests, conv / statistics tests), A/B on convs and C3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "test_conv_bf16 or epilogue or row_patch or concat" > gpurun_out/r4_rowp_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4_rowp_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_row_persist_ab.txt; : > $A
timeout -k 10 600 python tools/tile_ab.py 128x128x128x128x128 32x256x256x128x128 64x64x64x128x128 32x128x128x128x128 -- SDM_BF16_ROW_PERSIST=0 - SDM_BF16_ROW_PERSIST=0 - 2>/dev/null >> $A || exit 1
for v in 1 0 1 0; do
  echo "# SDM_BF16_ROW_PERSIST=$v" >> $A
  SDM_BF16_ROW_PERSIST=$v timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 10 2>/dev/null | grep forward >> $A || exit 1
  SDM_BF16_ROW_PERSIST=$v timeout -k 10 300 python tools/train_probe.py c5 bf16 6 2>/dev/null | grep "ms/step" >> $A || exit 1
done
cat $A
