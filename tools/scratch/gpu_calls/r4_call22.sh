# round 4: repeat the derived bf16 gate tests to see the spread of the worst HIP-vs-model / floor ratios (flake check)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r4_gate_repeat.log
for i in 1 2 3 4 5 6; do
  timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_configs.py -x -q -s -k "forward_bf16_vs_golden or standard_architecture_bf16 or grads_bf16_vs_reference or c3_ddim or training_step_128_bf16 or training_step_256_bf16" >> gpurun_out/r4_gate_repeat.log 2>&1 || { tail -30 gpurun_out/r4_gate_repeat.log; exit 1; }
done
grep "bf16 gate" gpurun_out/r4_gate_repeat.log | sed 's/(.*//' | sort | uniq -c | sort -k2 | cut -c1-150
