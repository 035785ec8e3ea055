# round 4: straight-line forward epilogue of igemm8: parity (bf16 suites), phase timing, A/B on convs and the C3 forward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r4_epi_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r4_epi_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
L=gpurun_out/r04_igemm8_timing_epi.txt; : > $L
SDM_HIP_LIB=$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_timing.so timeout -k 10 200 python tools/igemm8_timing.py 8 64 64 512 512 2>/dev/null >> $L || exit 1
SDM_HIP_LIB=$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_timing.so timeout -k 10 200 python tools/igemm8_timing.py 128 64 64 512 512 2>/dev/null >> $L || exit 1
echo "# same, SDM_IGEMM8_SIMPLE_EPI=0" >> $L
SDM_IGEMM8_SIMPLE_EPI=0 SDM_HIP_LIB=$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_timing.so timeout -k 10 200 python tools/igemm8_timing.py 128 64 64 512 512 2>/dev/null >> $L || exit 1
cat $L
A=gpurun_out/r04_igemm8_epi_ab.txt; : > $A
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 128x64x64x512x512 32x128x128x256x256 8x64x64x1024x1024 -- SDM_IGEMM8_SIMPLE_EPI=0 - SDM_IGEMM8_SIMPLE_EPI=0 - 2>/dev/null >> $A || exit 1
for v in 1 0 1 0; do
  echo "# SDM_IGEMM8_SIMPLE_EPI=$v" >> $A
  SDM_IGEMM8_SIMPLE_EPI=$v timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 10 2>/dev/null | grep forward >> $A || exit 1
done
cat $A
