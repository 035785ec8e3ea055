for s in "8 64 64 512 512" "8 128 128 256 256" "8 256 256 128 128" "8 32 32 1024 1024" "32 32 32 512 512"; do
  for m in fwd dgrad; do
    echo "$s $m: kcm   $(python tools/one_conv_bf16.py $s c3s1 20 $m 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*: //')"
    echo "$s $m: tapmaj $(SDM_BF16_KCM=0 python tools/one_conv_bf16.py $s c3s1 20 $m 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*: //')"
  done
done
