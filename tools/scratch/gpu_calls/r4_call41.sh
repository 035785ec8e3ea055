# round 4: the two slow deep-level data gradients of the C4 / C5 bf16 steps (M = 2048): tile / split choices
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r04_deep_dgrad.txt; : > $L
for f in "" "0,1" "1,1" "1,2" "1,4" "2,1"; do
  echo "# SDM_BF16_FORCE=$f" >> $L
  if [ -z "$f" ]; then E=""; else E="SDM_BF16_FORCE=$f"; fi
  env $E timeout -k 10 120 python tools/one_conv_bf16.py 32 8 8 1024 512 ct4 50 dgrad 2>/dev/null | tail -1 >> $L || exit 1
  env $E timeout -k 10 120 python tools/one_conv_bf16.py 32 16 16 512 512 c3s2 50 dgrad 2>/dev/null | tail -1 >> $L || exit 1
  env $E timeout -k 10 120 python tools/one_conv_bf16.py 32 8 8 1024 512 ct4 50 fwd 2>/dev/null | tail -1 >> $L || exit 1
  env $E timeout -k 10 120 python tools/one_conv_bf16.py 32 16 16 512 512 c3s2 50 fwd 2>/dev/null | tail -1 >> $L || exit 1
done
cat $L
