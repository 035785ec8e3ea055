# round 4: straight-line igemm8 epilogue also for table geometry (sub-pixel phases of ConvTranspose / stride-2 data gradient)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r4_epi_tests3.log 2>&1; rc=$?; tail -3 gpurun_out/r4_epi_tests3.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_epilogue_phase_ab.txt; : > $A
for v in 1 0 1 0; do
  echo "# SDM_IGEMM8_SIMPLE_EPI=$v" >> $A
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 200 python tools/one_conv_bf16.py 128 32 32 1024 256 ct4 30 fwd 2>/dev/null | tail -1 >> $A || exit 1
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 200 python tools/one_conv_bf16.py 128 16 16 1024 512 ct4 30 fwd 2>/dev/null | tail -1 >> $A || exit 1
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 200 python tools/one_conv_bf16.py 32 64 64 256 512 c3s2 30 dgrad 2>/dev/null | tail -1 >> $A || exit 1
done
cat $A
