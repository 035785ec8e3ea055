# Round-4 profile set (run on the GPU box from the repo root):  bash tools/profile_round4.sh [attn|sample|train|c5|c3|traffic|dump|pmc_attn|pmc_conv|all]...
# Results land under gpurun_out/; the summaries quoted in DESIGN.md are copied to profiles/r04_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
stats() {   # stats <outdir> <program args...>: rocprofv3 kernel-trace summary of one command
  out=$1; shift
  rm -rf gpurun_out/$out
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- "$@" > gpurun_out/$out.log 2>&1 || { echo "$out failed"; tail -5 gpurun_out/$out.log; return 1; }
  f=$(ls -t gpurun_out/$out/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/$out.kernel_stats.csv
  echo "== $out"; cut -c1-170 "$f" | head -${LINES_SHOWN:-14}
  rm -rf gpurun_out/$out
}
pmc() {     # pmc <outdir> <counter> <program args...>
  out=$1; ctr=$2; shift; shift
  rm -rf gpurun_out/$out
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/$out -- "$@" > gpurun_out/$out.log 2>&1 || { echo "$out failed"; tail -5 gpurun_out/$out.log; return 1; }
}
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
conv_pmc() { # tag kernel-substring "shape text" flops bytes  then one_conv_bf16 args
  tag=$1; sub=$2; shape=$3; fl=$4; by=$5; shift 5
  i=0; dirs=""
  for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf gpurun_out/pmc_${tag}_$i
    timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python tools/one_conv_bf16.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pmc $tag pass $i failed"; tail -5 gpurun_out/pmc_${tag}_$i.log; return 1; }
    dirs="$dirs gpurun_out/pmc_${tag}_$i"
  done
  python tools/pmc_fold.py "$sub" gpurun_out/r04_${tag}_pmc.json "$shape" $fl $by $dirs | tail -25
  rm -rf $dirs
}
for what in "$@"; do
  case $what in
    attn)    stats prof_attn_twopass python tools/attn_probe.py 8 4096 512 6 1 ;;
    sample)  stats prof_sample_n1_f32 python tools/sample_probe.py 1 100 f32 eager && stats prof_sample_n1_bf16 python tools/sample_probe.py 1 100 bf16 eager ;;
    train)   stats prof_train_step python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline ;;
    c5)      stats prof_c5_bf16_train python tools/train_probe.py c5 bf16 6 ;;
    c4)      stats prof_c4_bf16_train python tools/train_probe.py c4 bf16 6 ;;
    c3)      stats prof_c3_fwd_bf16 python tools/fwd_probe.py 128 128 bf16 10 4 ;;
    traffic) pmc prof_fetch FETCH_SIZE python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline && pmc prof_write WRITE_SIZE python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline && python tools/hbm_traffic.py $(ls -t gpurun_out/prof_fetch/*/*counter_collection.csv | head -1) $(ls -t gpurun_out/prof_write/*/*counter_collection.csv | head -1) gpurun_out/r04_hbm_traffic.json | tail -3 ;;
    dump)    SDM_PROF_DUMP=gpurun_out/r04_train_step_launches.csv timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-configs --no-cpu-baseline > gpurun_out/prof_dump.json 2> gpurun_out/prof_dump.err; wc -l gpurun_out/r04_train_step_launches.csv ;;
    pmc_attn) bash tools/pmc_attn.sh 1 twopass > gpurun_out/r4_pmc_attn1.txt 2>&1; tail -30 gpurun_out/r4_pmc_attn1.txt ;;
    pmc_conv) conv_pmc igemm8_fwd512 igemm8_bf16_kernel "forward Conv2d 512->512 k3 on 8 images of 64x64 (M=32768, N=512, K=512 x 9 taps), bf16, phase-interleaved 256x256 tile" 154618822656 71827456 8 64 64 512 512 c3s1 4 fwd ;;
    pmc_conv_tapin) SDM_BF16_TAPIN=1 conv_pmc igemm8_tapin_fwd512 igemm8_bf16_kernel "forward Conv2d 512->512 k3 on 8 images of 64x64 (M=32768, N=512, K=512 x 9 taps), bf16, phase-interleaved 256x256 tile, taps innermost in the K walk" 154618822656 71827456 8 64 64 512 512 c3s1 4 fwd ;;
    pmc_wgrad) conv_pmc wgrad8_512 wgrad8x3_bf16_kernel "weight gradient of Conv2d 512->512 k3 on 8 images of 64x64 (M=N=512, 32768 pixels), bf16, three taps per workgroup, phase-interleaved" 154618822656 76546048 8 64 64 512 512 c3s1 4 wgrad && conv_pmc wgrad8_128 wgrad8x3_bf16_kernel "weight gradient of Conv2d 128->128 k3 on 8 images of 256x256 (M=N=128, 524288 pixels), bf16, three taps per workgroup, phase-interleaved" 154618822656 269025280 8 256 256 128 128 c3s1 4 wgrad ;;
    *) echo "unknown $what" ;;
  esac
done
