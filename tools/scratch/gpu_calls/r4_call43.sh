# round 4: the deep-level data gradients with COLD operands (1 GiB of writes between calls): tile / split / ring choices
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r04_deep_dgrad_cold.txt; : > $L
for f in "" "0,1" "1,2" "1,4" "1,8" "2,1"; do
  echo "# SDM_BF16_FORCE=$f" >> $L
  if [ -z "$f" ]; then E="X=1"; else E="SDM_BF16_FORCE=$f"; fi
  env ONE_CONV_COLD=1 $E timeout -k 10 120 python tools/one_conv_bf16.py 32 8 8 1024 512 ct4 20 dgrad 2>/dev/null | tail -1 >> $L || exit 1
  env ONE_CONV_COLD=1 $E timeout -k 10 120 python tools/one_conv_bf16.py 32 8 8 512 512 c3s1 20 dgrad 2>/dev/null | tail -1 >> $L || exit 1
  env ONE_CONV_COLD=1 $E timeout -k 10 120 python tools/one_conv_bf16.py 32 8 8 1024 1024 c3s1 20 dgrad 2>/dev/null | tail -1 >> $L || exit 1
done
cat $L
