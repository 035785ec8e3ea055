# A/B of the split-K finishing-launch charge in the planners on the N=1 / N=4 sampling step and the batch-4 training step
for sc in 6000 12000 20000 30000; do
  echo "== SDM_IGEMM_SPLIT_COST=$sc"; SDM_IGEMM_SPLIT_COST=$sc python tools/sample_timing.py 2>&1 | grep "graph=True" | head -2 | tr "\n" " "; echo
  SDM_IGEMM_SPLIT_COST=$sc python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train ms', round(d['ms_per_step'],3), 'launches', d['launches_per_step'])"
done
for te in 0 2.5 5 8; do
  echo "== SDM_BF16_SPLIT_COST=$te"; SDM_BF16_SPLIT_COST=$te python tools/sample_timing.py 2>&1 | grep "graph=True" | tail -2 | tr "\n" " "; echo
done
echo "== FUSE_GN_BF16=1"; SDM_FUSE_GN_BF16=1 python tools/sample_timing.py 2>&1 | grep "graph=True" | tail -2 | tr "\n" " "; echo
