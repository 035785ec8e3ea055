# round 4: after removing the in-kernel split-K code and the fp32 wgrad tap co-location: parity of the touched kernels, headline A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -k 'conv_fwd_dgrad_wgrad or concat or unet_grads_golden or standard_architecture or test_conv_bf16 or forward_bf16' > gpurun_out/r4_t25.log 2>&1 || { tail -40 gpurun_out/r4_t25.log; exit 1; }
tail -2 gpurun_out/r4_t25.log
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'avg_launch_us', round(r['avg_launch_ms']*1e3,2), {k:round(v['ms_per_step'],3) for k,v in r['by_class'].items() if v['ms_per_step']>0})"
done
