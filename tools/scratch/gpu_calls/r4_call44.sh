# round 4: deep-level data gradient with cold caches AND cold address translation (one word per 4 KiB page of 32 / 96 GiB touched between calls)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r04_deep_dgrad_tlb.txt; : > $L
for g in 0 32 96; do
  echo "# ONE_CONV_TLB=$g GiB" >> $L
  ONE_CONV_COLD=1 ONE_CONV_TLB=$g timeout -k 10 200 python tools/one_conv_bf16.py 32 8 8 1024 512 ct4 10 dgrad 2>/dev/null | tail -1 >> $L || exit 1
  ONE_CONV_COLD=1 ONE_CONV_TLB=$g timeout -k 10 200 python tools/one_conv_bf16.py 32 8 8 512 512 c3s1 10 dgrad 2>/dev/null | tail -1 >> $L || exit 1
  ONE_CONV_COLD=1 ONE_CONV_TLB=$g timeout -k 10 200 python tools/one_conv_bf16.py 32 16 16 1024 1024 c3s1 10 dgrad 2>/dev/null | tail -1 >> $L || exit 1
done
cat $L
