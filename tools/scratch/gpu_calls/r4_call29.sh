# round 4: re-sweep of the fp32 tile / split choices on the headline's shapes with today's kernels (planner calibration check)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python tools/sweep_igemm.py > gpurun_out/r4_igemm_sweep.log 2>&1; cut -c1-190 gpurun_out/r4_igemm_sweep.log
