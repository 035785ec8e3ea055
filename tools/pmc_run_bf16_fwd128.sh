cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
i=0
for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_fw128_$i -- python tools/one_conv_bf16.py 8 256 256 128 128 c3s1 4 fwd > gpurun_out/pmc_fw128_$i.log 2>&1 || exit 1
done
python tools/one_conv_bf16.py 8 256 256 128 128 c3s1 20 fwd 2>&1 | grep "per call"
