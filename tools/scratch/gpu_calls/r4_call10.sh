# round 4: derived bf16 gates on the GPU; attention counters; kernel totals of the bf16 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_configs.py tests/test_gpu_objectives.py -x -q -s -k "forward_bf16_vs_golden or standard_architecture_bf16 or grads_bf16_vs_reference or c3_ddim or training_step_128_bf16 or training_step_256_bf16 or checkpoint_resume" > gpurun_out/r4_t10.log 2>&1 || { tail -60 gpurun_out/r4_t10.log; exit 1; }
grep "bf16 gate\|passed\|failed" gpurun_out/r4_t10.log
bash tools/profile_round4.sh pmc_attn attn c5 c3 > gpurun_out/r4_prof_b.txt 2>&1; tail -90 gpurun_out/r4_prof_b.txt | cut -c1-180
