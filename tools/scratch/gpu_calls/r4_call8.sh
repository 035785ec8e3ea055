# round 4: phase-interleaved three-tap weight gradient: forced-kernel tests, A/B timing; Adam copies test; tile A/B; bench; probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_train.py -x -q -k "adam_writes or trainer_bf16" > gpurun_out/r4_t8.log 2>&1 || { tail -60 gpurun_out/r4_t8.log; exit 1; }
tail -3 gpurun_out/r4_t8.log
timeout -k 10 600 python tools/wgrad_ab.py 8x256x256x128x128 8x128x128x256x256 8x64x64x512x512 32x32x32x512x512 8x64x64x1024x512 32x16x16x1024x1024 -- SDM_WGRAD8=0 - SDM_WGRAD3=0,SDM_WGRAD8=0 > gpurun_out/r4_wgrad_ab.log 2>&1 || { tail -20 gpurun_out/r4_wgrad_ab.log; exit 1; }
cat gpurun_out/r4_wgrad_ab.log
timeout -k 10 600 python tools/tile_ab.py 8x64x64x512x512 8x128x128x256x256 32x128x128x128x128 8x32x32x1024x1024 8x256x256x128x128 -- - SDM_BF16_8PH=0 > gpurun_out/r4_tile_ab.log 2>&1 || { tail -20 gpurun_out/r4_tile_ab.log; exit 1; }
cat gpurun_out/r4_tile_ab.log
timeout -k 10 1000 python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err || { tail -20 gpurun_out/r4_bench1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_bench1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'))
for k,v in d.get('configs',{}).items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','steps_per_s','launches_per_step','bf16','f32','per_gpu_batch')})
P
timeout -k 10 900 python tools/bf16_oracle_probe.py > gpurun_out/r4_bf16_oracle_probe.log 2>&1; tail -40 gpurun_out/r4_bf16_oracle_probe.log
