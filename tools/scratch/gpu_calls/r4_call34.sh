# round 4: branch-free stem kernel (4 channel blocks per column, activation as a template parameter)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "head_conv or stem_conv" > gpurun_out/r4_edge_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r4_edge_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
L=gpurun_out/r4_edge_probe2.log; : > $L
timeout -k 10 300 python tools/edge_probe.py both >> $L 2>&1 || exit 1
timeout -k 10 300 python tools/edge_probe.py both 64 64 >> $L 2>&1 || exit 1
timeout -k 10 300 python tools/edge_probe.py both 64 256 >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L | cut -c1-200
