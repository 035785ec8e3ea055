# round 4: is the fp32 64x64 class slower than in round 3?  current library against a hybrid with round 3's sdm_igemm.hip, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for lib in "" "$GRAFT_REPO_ROOT/simple-diffusion-model_amd/sdm_hip/libsdm_hip_r03igemm.so"; do
  echo "== lib=${lib:-current}"
  SDM_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'avg_launch_us', round(r['avg_launch_ms']*1e3,2), {k:round(v['ms_per_step'],3) for k,v in r['by_class'].items() if v['ms_per_step']>0})"
done; done
