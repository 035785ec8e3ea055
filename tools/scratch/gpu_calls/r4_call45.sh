# round 4: phase timing inside igemm8_bf16_kernel (timing build: libsdm_hip_timing.so = product objects + sdm_igemm8.hip with -DSDM_IGEMM8_TIMING)
# build (in the CPU container):  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DSDM_IGEMM8_TIMING -c csrc/sdm_igemm8.hip -o /tmp/t.o
#                                hipcc --offload-arch=gfx950 -shared -fPIC -o sdm_hip/libsdm_hip_timing.so <product objects with /tmp/t.o for sdm_igemm8.o> -ldl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SDM_HIP_LIB=$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_timing.so
L=gpurun_out/r04_igemm8_timing.txt; : > $L
timeout -k 10 200 python tools/igemm8_timing.py 8 64 64 512 512 2>/dev/null >> $L || exit 1
timeout -k 10 200 python tools/igemm8_timing.py 128 64 64 512 512 2>/dev/null >> $L || exit 1
timeout -k 10 200 python tools/igemm8_timing.py 8 64 64 1024 1024 2>/dev/null >> $L || exit 1
cat $L
