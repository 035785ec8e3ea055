# round 4: the vendor's shape of kernel (4 waves of 128x128, one barrier per K-tile) as a microbenchmark, against the 8-wave loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== gemm4w conv-like 8x64x64 512->512 (9 taps)"; timeout -k 10 120 tools/scratch/micro/gemm4w 32768 512 512 9 64 &&
echo "== gemm4w plain GEMM 32768x512x4608"; timeout -k 10 120 tools/scratch/micro/gemm4w 32768 512 4608 1 64 &&
echo "== gemm4w 4096^3"; timeout -k 10 120 tools/scratch/micro/gemm4w 4096 4096 4096 1 64 &&
echo "== gemm8ph conv-like 8x64x64 512->512 (9 taps)"; timeout -k 10 120 tools/scratch/micro/gemm8ph 32768 512 512 9 64 &&
echo "== gemm8ph plain GEMM 32768x512x4608"; timeout -k 10 120 tools/scratch/micro/gemm8ph 32768 512 4608 1 64 &&
echo "== vendor GEMM"; timeout -k 10 200 python tools/vendor_gemm.py 32768x512x4608 4096x4096x4096
} > gpurun_out/r4_gemm4w.log 2>&1
cat gpurun_out/r4_gemm4w.log | cut -c1-170
