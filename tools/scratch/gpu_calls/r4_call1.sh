# round 4, first GPU call: the phase-interleaved microbenchmark against the product kernel and the vendor GEMM on ONE box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=tools/scratch/micro/gemm8ph
{
echo "== gemm8ph conv-like 8x64x64 512->512 (9 taps)"; timeout -k 10 120 $B 32768 512 512 9 64 &&
echo "== gemm8ph plain GEMM 32768x512x4608"; timeout -k 10 120 $B 32768 512 4608 1 64 &&
echo "== gemm8ph 4096^3"; timeout -k 10 120 $B 4096 4096 4096 1 64 &&
echo "== gemm8ph 8192x1024x9216 (1024 ch at 32x32)"; timeout -k 10 120 $B 8192 1024 1024 9 32 &&
echo "== gemm8ph 131072x256x2304 (256 ch at 128x128)"; timeout -k 10 120 $B 131072 256 256 9 128 &&
echo "== product kernel (tilecmp)"; timeout -k 10 300 python tools/scratch/tilecmp.py 8x64x64x512x512 8x128x128x256x256 8x32x32x1024x1024 &&
echo "== vendor GEMM"; timeout -k 10 200 python tools/vendor_gemm.py 32768x512x4608 4096x4096x4096 8192x1024x9216 131072x256x2304
} > gpurun_out/r4_call1.log 2>&1
rc=$?
tail -60 gpurun_out/r4_call1.log
[ $rc -eq 0 ] || exit $rc
# vendor kernel names + PMC passes of the 512-channel shape
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vendor_trace -- python tools/vendor_gemm.py 32768x512x4608 8192x1024x9216 131072x256x2304 > gpurun_out/vendor_trace.log 2>&1 || exit 1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
i=0
for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_vendor_$i -- python tools/vendor_gemm.py 32768x512x4608 > gpurun_out/pmc_vendor_$i.log 2>&1 || exit 1
done
i=0
for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_g8_$i -- $B 32768 512 512 9 64 5 > gpurun_out/pmc_g8_$i.log 2>&1 || exit 1
done
echo done
