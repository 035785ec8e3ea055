# Round-3 profiling passes (run on the GPU box from the repo root):  bash tools/profile_round3.sh [what...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stats() {   # stats <outdir> <program args...>: rocprofv3 kernel-trace summary of one command
  out=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- "$@" > gpurun_out/$out.log 2>&1 || { echo "$out failed"; tail -5 gpurun_out/$out.log; return 1; }
  f=$(find gpurun_out/$out -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/$out.kernel_stats.csv
  echo "== $out"; cut -c1-170 "$f" | head -${LINES_SHOWN:-14}
}
for what in "$@"; do
  case $what in
    attn)   stats prof_attn_twopass python tools/attn_probe.py 8 4096 512 6 1 && stats prof_attn_scores python tools/attn_probe.py 8 4096 512 6 0 ;;
    sample) stats prof_sample_n1_f32 python tools/sample_probe.py 1 100 f32 eager && stats prof_sample_n1_bf16 python tools/sample_probe.py 1 100 bf16 eager ;;
    train)  stats prof_train_step python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline ;;
    *) echo "unknown $what" ;;
  esac
done
