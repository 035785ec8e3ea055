cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
i=0
for grp in "$G1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_attn_$i -- python tools/attn_probe.py 8 4096 512 3 > gpurun_out/pmc_attn_$i.log 2>&1 || exit 1
done
tail -1 gpurun_out/pmc_attn_1.log
