# round 4: per-launch dump of the N=1 fp32 forward (64x64): which contraction launches stream weights slowly
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SDM_PROF_DUMP=gpurun_out/r04_n1_fwd_launches.csv timeout -k 10 300 python tools/fwd_launch_dump.py 1 2>/dev/null | tail -1 | cut -c1-400 &&
SDM_PROF_DUMP=gpurun_out/r04_n1_fwd_launches_bf16.csv timeout -k 10 300 python tools/fwd_launch_dump_bf16.py 64 1 0 1 2>/dev/null | tail -1 | cut -c1-400
