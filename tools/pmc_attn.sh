# PMC passes of the bf16 attention core at S=4096, d=512, 8 images (C5's largest attention): bash tools/pmc_attn.sh <twopass 0|1> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
tp=${1:-1}; tag=${2:-twopass}
i=0
for grp in "$G1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_attn_${tag}_$i
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_attn_${tag}_$i -- python tools/attn_probe.py 8 4096 512 3 $tp > gpurun_out/pmc_attn_${tag}_$i.log 2>&1 || exit 1
done
tail -1 gpurun_out/pmc_attn_${tag}_1.log
per=5; [ "$tp" = "0" ] && per=4
python tools/pmc_attn_fold.py gpurun_out/attention_bf16_pmc_${tag}.json 8 4096 512 $per gpurun_out/pmc_attn_${tag}_1 gpurun_out/pmc_attn_${tag}_2 gpurun_out/pmc_attn_${tag}_3
