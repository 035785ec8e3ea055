# Round-3 profile set (run on the GPU box from the repo root):  bash tools/profile_round3.sh [attn|sample|train|traffic|dump|pmc|all]...
# Results land under gpurun_out/; the summaries quoted in DESIGN.md are copied to profiles/r03_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stats() {   # stats <outdir> <program args...>: rocprofv3 kernel-trace summary of one command
  out=$1; shift
  rm -rf gpurun_out/$out
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- "$@" > gpurun_out/$out.log 2>&1 || { echo "$out failed"; tail -5 gpurun_out/$out.log; return 1; }
  f=$(ls -t gpurun_out/$out/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/$out.kernel_stats.csv
  echo "== $out"; cut -c1-170 "$f" | head -${LINES_SHOWN:-14}
}
pmc() {     # pmc <outdir> <counter> <program args...>
  out=$1; ctr=$2; shift; shift
  rm -rf gpurun_out/$out
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/$out -- "$@" > gpurun_out/$out.log 2>&1 || { echo "$out failed"; tail -5 gpurun_out/$out.log; return 1; }
}
for what in "$@"; do
  case $what in
    attn)    stats prof_attn_twopass python tools/attn_probe.py 8 4096 512 6 1 && stats prof_attn_scores python tools/attn_probe.py 8 4096 512 6 0 ;;
    sample)  stats prof_sample_n1_f32 python tools/sample_probe.py 1 100 f32 eager && stats prof_sample_n1_bf16 python tools/sample_probe.py 1 100 bf16 eager ;;
    train)   stats prof_train_step python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline ;;
    traffic) pmc prof_fetch FETCH_SIZE python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline && pmc prof_write WRITE_SIZE python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline && python tools/hbm_traffic.py $(ls -t gpurun_out/prof_fetch/*/*counter_collection.csv | head -1) $(ls -t gpurun_out/prof_write/*/*counter_collection.csv | head -1) gpurun_out/r03_hbm_traffic.json | tail -3 ;;
    dump)    SDM_PROF_DUMP=gpurun_out/r03_train_step_launches.csv timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-configs --no-cpu-baseline > gpurun_out/prof_dump.json 2> gpurun_out/prof_dump.err; wc -l gpurun_out/r03_train_step_launches.csv ;;
    pmc)     bash tools/pmc_attn.sh 1 twopass > gpurun_out/r3_pmc_attn1.txt 2>&1; bash tools/pmc_attn.sh 0 scores > gpurun_out/r3_pmc_attn0.txt 2>&1; tail -12 gpurun_out/r3_pmc_attn1.txt ;;
    all)     bash tools/profile_round3.sh train sample attn traffic dump pmc ;;
    *) echo "unknown $what" ;;
  esac
done
