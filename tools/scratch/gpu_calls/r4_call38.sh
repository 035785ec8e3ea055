# round 4: per-launch dumps of the C5 and C4 bf16 training steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SDM_PROF_DUMP=gpurun_out/r04_c5_bf16_launches.csv timeout -k 10 300 python tools/train_launch_dump.py c5 bf16 2>/dev/null | tail -1 | cut -c1-400 &&
SDM_PROF_DUMP=gpurun_out/r04_c4_bf16_launches.csv timeout -k 10 300 python tools/train_launch_dump.py c4 bf16 2>/dev/null | tail -1 | cut -c1-400
