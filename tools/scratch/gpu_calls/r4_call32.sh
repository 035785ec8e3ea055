# round 4: edge kernels with deeper prefetch: parity, then timing of stem (8 / 4 channel blocks per column) and head
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "head_conv or stem_conv" > gpurun_out/r4_edge_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r4_edge_tests.log | cut -c1-300
[ $rc -eq 0 ] && timeout -k 10 300 python tools/edge_probe.py > gpurun_out/r4_edge_probe.log 2>&1 && SDM_STEM_NB=4 timeout -k 10 300 python tools/edge_probe.py stem >> gpurun_out/r4_edge_probe.log 2>&1; cat gpurun_out/r4_edge_probe.log | cut -c1-200
