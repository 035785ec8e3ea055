"""Where a workgroup of igemm8_bf16_kernel spends its time (tuning aid).  Needs the timing build of the library:
    make -C simple-diffusion-model_amd/csrc && (see tools/scratch/gpu_calls/r4_call45.sh for the two-line build of
    libsdm_hip_timing.so: sdm_igemm8.hip compiled with -DSDM_IGEMM8_TIMING, linked with the product objects)
    SDM_HIP_LIB=.../libsdm_hip_timing.so python tools/igemm8_timing.py n h w cin cout [kind=c3s1]
Stamps (s_memrealtime, 100 MHz) of wave 0 of every workgroup: entry, prologue done, first operands landed, K loop done, epilogue done."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.custom_layers import RunCtx, TapConv  # noqa: E402
from sdm_hip import lib, ops  # noqa: E402

n, h, w, cin, cout = map(int, sys.argv[1:6])
kind = sys.argv[6] if len(sys.argv) > 6 else "c3s1"
ROW = os.environ.get("TIMING_KERNEL") == "row"        # the row-patch kernel's stamps (128x128 tiles) instead of igemm8's
conv = TapConv(cin, cout, kind).cuda()
ctx = RunCtx(False, bf16=True)
x = ops.Act(torch.randn(n * h * w, cin, device="cuda").to(torch.bfloat16), n, h, w)
wt = conv.taps_for(ctx)
for _ in range(3):
    y = ops.conv_fwd(x, wt, conv.bias, kind, ops.ACT_SILU)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
y = ops.conv_fwd(x, wt, conv.bias, kind, ops.ACT_SILU)
b.record()
torch.cuda.synchronize()
if ROW:
    tiles = ((n * h * w + 127) // 128) * ((cout + 127) // 128)
    nwg = min(tiles, 65536)
    buf = (C.c_ulonglong * (nwg * 4))()
    f = lib.raw("sdm_debug_row_times")
    f.restype = C.c_int
    assert f(buf, nwg) == 0
    t4 = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.int64)
    t = np.stack([t4[:, 0], t4[:, 1], t4[:, 1], t4[:, 2], t4[:, 3]], axis=1)     # (no "first operands" stamp: folded into the loop)
else:
    tiles = ((n * h * w + 255) // 256) * ((cout + 255) // 256)
    nwg = min(tiles, 16384)
    buf = (C.c_ulonglong * (nwg * 6))()
    f = lib.raw("sdm_debug_igemm8_times")
    f.restype = C.c_int
    assert f(buf, nwg) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 6).astype(np.int64)[:, :5]
t0 = t[:, 0].min()
us = (t - t0) / 100.0                      # 100 MHz -> us since the first workgroup's entry
seg = np.diff(us, axis=1)
taps = 16 if kind == "ct4" else 9
kt = taps * cin // 64
print(f"{kind} n{n} {h}x{w} {cin}->{cout}: {tiles} tiles of {'128x128 (row-patch kernel, 2 workgroups per CU)' if ROW else '256x256'}, {kt} K-tiles each; launch {a.elapsed_time(b) * 1e3:.1f} us (event pair)")
print(f"  kernel span (first entry -> last epilogue end): {us[:, 4].max():.1f} us")
names = ["prologue (tables, first 7 stagings issued)", "wait for the first operands", "K loop", "epilogue"]
for i, nm in enumerate(names):
    print(f"  {nm:46s} mean {seg[:, i].mean():7.2f} us   p10 {np.percentile(seg[:, i], 10):7.2f}   p90 {np.percentile(seg[:, i], 90):7.2f}")
print(f"  per K-tile in the loop: {seg[:, 2].mean() / kt:.3f} us;  workgroup total mean {np.sum(seg, axis=1).mean():.2f} us")
starts = np.sort(us[:, 0])
print(f"  entry times: first 256 workgroups within {starts[min(255, nwg - 1)]:.2f} us; "
      f"rounds (tiles / {'512 slots' if ROW else '256 CUs'}): {tiles / (512 if ROW else 256):.1f}; sum of workgroup times / {'512' if ROW else '256'} = {np.sum(seg) / (512 if ROW else 256):.1f} us")
# idle between consecutive workgroups on a CU cannot be seen directly; the difference between the span and the per-CU busy sum is it
