# round 4: straight-line epilogues in igemm8 (+ data-gradient extras), the row-patch kernel and the generic bf16 tiles: parity + A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r4_epi_tests2.log 2>&1; rc=$?; tail -4 gpurun_out/r4_epi_tests2.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_epilogue_ab.txt; : > $A
for v in 1 0 1 0; do
  echo "# SDM_IGEMM8_SIMPLE_EPI=$v" >> $A
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 10 2>/dev/null | grep forward >> $A || exit 1
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 300 python tools/train_probe.py c4 bf16 6 2>/dev/null | grep "ms/step" >> $A || exit 1
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 300 python tools/train_probe.py c5 bf16 6 2>/dev/null | grep "ms/step" >> $A || exit 1
done
cat $A
