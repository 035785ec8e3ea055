#!/bin/bash
# wgrad tile comparison: default plan with the 256 tile disabled (T256 huge) vs forced 256 at several splits
for s in "8 64 64 512 512" "8 128 128 256 256" "8 32 32 1024 1024" "32 32 32 512 512" "32 64 64 256 256" "32 16 16 1024 1024"; do
  echo "$s: no256: $(SDM_WGRAD_BF16_T256=1000 python tools/one_conv_bf16.py $s c3s1 20 wgrad 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*: //')"
  echo "$s: plan  : $(python tools/one_conv_bf16.py $s c3s1 20 wgrad 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*: //')"
  for sk in 4 7 14 28; do
    echo "$s: 256,$sk: $(SDM_WGRAD_BF16_FORCE=256,$sk python tools/one_conv_bf16.py $s c3s1 20 wgrad 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*: //')"
  done
done
