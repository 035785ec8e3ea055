cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run() { # tag mode n h w cin cout
  tag=$1; shift
  i=0
  for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python tools/one_conv_bf16.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
  done
  tail -1 gpurun_out/pmc_${tag}_1.log
}
run w3_128 8 256 256 128 128 c3s1 4 wgrad && run row128 8 256 256 128 128 c3s1 4 fwd
