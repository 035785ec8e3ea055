# round 4: counters of the two phase-interleaved kernels in the product, bf16-oracle probe, bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round4.sh pmc_conv pmc_wgrad > gpurun_out/r4_pmc_a.txt 2>&1; tail -80 gpurun_out/r4_pmc_a.txt | cut -c1-200
timeout -k 10 900 python tools/bf16_oracle_probe.py > gpurun_out/r4_bf16_oracle_probe.log 2>&1; tail -45 gpurun_out/r4_bf16_oracle_probe.log
timeout -k 10 1000 python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err || { tail -20 gpurun_out/r4_bench1.err; exit 1; }
tail -5 gpurun_out/r4_bench1.err
