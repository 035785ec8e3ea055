# round 4: edge convolutions of the bf16 path on the matrix cores -- parity tests, then the C3 forward profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "head_conv or stem_conv or forward_bf16 or standard_arch" > gpurun_out/r4_edge_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r4_edge_tests.log | cut -c1-300
[ $rc -eq 0 ] && timeout -k 10 600 bash tools/profile_round4.sh c3 > gpurun_out/r4_c3_prof.log 2>&1 && tail -5 gpurun_out/r4_c3_prof.log | cut -c1-300
