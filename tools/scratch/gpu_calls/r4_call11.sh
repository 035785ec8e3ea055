# round 4: what bounds wgrad8x3 (component-removal timing), in-kernel split-K A/B on the headline and the N=1 sampler
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/wgrad_ab.py 8x64x64x512x512 8x256x256x128x128 -- - SDM_WGRAD8_DBG=1 SDM_WGRAD8_DBG=2 SDM_WGRAD8_DBG=3 SDM_WGRAD8_DBG=4 SDM_WGRAD8_DBG=8 SDM_WGRAD8_DBG=9 SDM_WGRAD8_DBG=10 SDM_WGRAD8_DBG=11 > gpurun_out/r4_wgrad8_dbg.log 2>&1
cut -c1-140 gpurun_out/r4_wgrad8_dbg.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "split_k_is_finished or conv_fwd_dgrad or sampler_traj or standard_architecture" > gpurun_out/r4_t11.log 2>&1 || { tail -40 gpurun_out/r4_t11.log; exit 1; }
tail -2 gpurun_out/r4_t11.log
for v in 1 0; do
  echo "== SDM_SPLITK_INKERNEL=$v"
  SDM_SPLITK_INKERNEL=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms/step', d['ms_per_step'], 'launches', d.get('launches_per_step'), 'host', d.get('host_enqueue_ms_per_step'))"
  SDM_SPLITK_INKERNEL=$v timeout -k 10 300 python tools/sample_probe.py 1 300 f32 graph 2>&1 | tail -1
  SDM_SPLITK_INKERNEL=$v timeout -k 10 300 python tools/sample_probe.py 1 300 bf16 graph 2>&1 | tail -1
done
