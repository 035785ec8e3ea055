# round 4: phases of the fp32 contraction kernel's workgroups on the headline's conv shapes (batch 4, 64x64)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SDM_HIP_LIB=$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_timing.so
L=gpurun_out/r04_f32_timing.txt; : > $L
for shp in "4 64 64 128 128" "4 32 32 256 256" "4 16 16 512 512" "4 8 8 512 512" "4 16 16 1024 512" "4 32 32 512 256"; do
  timeout -k 10 120 python tools/f32_timing.py $shp 2>/dev/null >> $L || exit 1
done
cat $L
