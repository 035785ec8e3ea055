# round 4: 256x128 row-patch tile with a 3- / 4-slot weight ring (counted waits, asm fragment reads): parity + A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=gpurun_out/r04_row256_deep_ab.txt; : > $A
timeout -k 10 600 python tools/tile_ab.py 128x128x128x128x128 32x256x256x128x128 64x64x64x128x128 128x128x128x64x128 -- - SDM_BF16_ROW256=1 SDM_BF16_ROW256=3 SDM_BF16_ROW256=4 - SDM_BF16_ROW256=3 SDM_BF16_ROW256=4 2>/dev/null >> $A || exit 1
cat $A
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "row_patch" > gpurun_out/r4_row256_deep_tests.log 2>&1; tail -3 gpurun_out/r4_row256_deep_tests.log | cut -c1-300
