# round 4: phase timing of the row-patch kernel's workgroups (128-channel convs) + PMC refresh of igemm8 after the epilogue change
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SDM_HIP_LIB=$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_timing.so
L=gpurun_out/r04_row_timing.txt; : > $L
TIMING_KERNEL=row timeout -k 10 200 python tools/igemm8_timing.py 128 128 128 128 128 2>/dev/null >> $L || exit 1
TIMING_KERNEL=row timeout -k 10 200 python tools/igemm8_timing.py 32 64 64 128 128 2>/dev/null >> $L || exit 1
TIMING_KERNEL=row timeout -k 10 200 python tools/igemm8_timing.py 12 64 64 256 128 2>/dev/null >> $L || exit 1
cat $L
unset SDM_HIP_LIB
timeout -k 10 900 bash tools/profile_round4.sh pmc_conv > gpurun_out/r4_pmc_conv2.log 2>&1; tail -12 gpurun_out/r4_pmc_conv2.log | cut -c1-220
