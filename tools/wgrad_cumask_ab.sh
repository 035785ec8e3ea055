#!/bin/bash
# A/B of a CU-masked weight-gradient stream (SDM_WGRAD_CUS, ops._SideStream._masked_stream): headline fp32 step and C4 bf16 step.
#   bash tools/wgrad_cumask_ab.sh > gpurun_out/wgrad_cumask.txt
for n in 0 64 128 192 0; do
  echo "== SDM_WGRAD_CUS=$n"
  SDM_WGRAD_CUS=$n python tools/train_probe.py c2 f32 40 2>&1 | grep -v amdgpu.ids || exit 1
  SDM_WGRAD_CUS=$n python tools/train_probe.py c4 bf16 6 2>&1 | grep -v amdgpu.ids || exit 1
done
