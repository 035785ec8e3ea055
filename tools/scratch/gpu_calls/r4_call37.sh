# round 4: same-box A/B of the edge kernels on the C3 forward (128 images of 128x128, bf16) and the SR forward (8 x 256x256)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r04_edge_ab.txt; : > $L
for rep in 1 2; do
  echo "# edge kernels ON (default)" >> $L
  timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 10 2>/dev/null | grep forward >> $L || exit 1
  echo "# edge kernels OFF (SDM_HEAD_STRIP=0 SDM_STEM_MFMA=0)" >> $L
  SDM_HEAD_STRIP=0 SDM_STEM_MFMA=0 timeout -k 10 300 python tools/fwd_probe.py 128 128 bf16 10 10 2>/dev/null | grep forward >> $L || exit 1
done
echo "# edge probe (standalone kernels, 20 launches each)" >> $L
timeout -k 10 300 python tools/edge_probe.py both 2>/dev/null | grep -v amdgpu >> $L
SDM_HEAD_STRIP=0 SDM_STEM_MFMA=0 timeout -k 10 300 python tools/edge_probe.py both 2>/dev/null | grep -v amdgpu | sed 's/^/OFF: /' >> $L
cat $L
