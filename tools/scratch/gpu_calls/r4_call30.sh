# round 4: per-launch dump of three C3 forwards (bf16, 128 images of 128x128): where are the slow 128x128-tile launches?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SDM_PROF_DUMP=gpurun_out/r04_c3_fwd_launches.csv timeout -k 10 600 python tools/fwd_launch_dump_bf16.py 128 128 10 3 > gpurun_out/r4_c3_dump.log 2>&1 && tail -3 gpurun_out/r4_c3_dump.log | cut -c1-400
