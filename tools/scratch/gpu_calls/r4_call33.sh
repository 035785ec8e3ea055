# round 4: packed-rows head kernel (variants) and stem component-removal experiments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "head_conv or stem_conv" > gpurun_out/r4_edge_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r4_edge_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
L=gpurun_out/r4_edge_probe.log; : > $L
for v in 34 43 44; do echo "SDM_HEAD_VARIANT=$v" >> $L; SDM_HEAD_VARIANT=$v timeout -k 10 300 python tools/edge_probe.py head >> $L 2>&1 || exit 1; done
echo "SDM_HEAD_PACKED=0" >> $L; SDM_HEAD_PACKED=0 timeout -k 10 300 python tools/edge_probe.py head >> $L 2>&1 || exit 1
for d in 0 1 2 3; do echo "SDM_STEM_DBG=$d" >> $L; SDM_STEM_DBG=$d timeout -k 10 300 python tools/edge_probe.py stem >> $L 2>&1 || exit 1; done
grep -v amdgpu.ids $L | cut -c1-200
