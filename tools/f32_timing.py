"""Phases of the fp32 contraction kernel's workgroups (igemm_glds_kernel) on one conv, as planned by the library (tuning aid;
needs the timing build of the library, see tools/igemm8_timing.py):
    SDM_HIP_LIB=.../libsdm_hip_timing.so python tools/f32_timing.py n h w cin cout [kind=c3s1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.custom_layers import TapConv  # noqa: E402
from sdm_hip import lib, ops  # noqa: E402

n, h, w, cin, cout = map(int, sys.argv[1:6])
kind = sys.argv[6] if len(sys.argv) > 6 else "c3s1"
conv = TapConv(cin, cout, kind).cuda()
x = ops.Act(torch.randn(n * h * w, cin, device="cuda"), n, h, w)
for _ in range(3):
    y = ops.conv_fwd(x, conv.taps(), conv.bias, kind, ops.ACT_SILU)
torch.cuda.synchronize()
clr = lib.raw("sdm_debug_f32_clear")
clr.restype = C.c_int
assert clr() == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
y = ops.conv_fwd(x, conv.taps(), conv.bias, kind, ops.ACT_SILU)
b.record()
torch.cuda.synchronize()
N = 65536
buf = (C.c_ulonglong * (N * 4))()
f = lib.raw("sdm_debug_f32_times")
f.restype = C.c_int
assert f(buf, N) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(N, 4).astype(np.int64)
t = t[t[:, 0] > 0]
nwg = len(t)
us = (t - t[:, 0].min()) / 100.0
seg = np.diff(us, axis=1)
print(f"fp32 {kind} n{n} {h}x{w} {cin}->{cout}: {nwg} workgroups of the contraction kernel (the finishing launch of a split is not in here); "
      f"event pair {a.elapsed_time(b) * 1e3:.1f} us; kernel span {us[:, 3].max():.1f} us")
for i, nm in enumerate(["prologue (tables, row coordinates)", "K loop (incl. first operands)", "epilogue (stores or split-K atomics)"]):
    print(f"  {nm:40s} mean {seg[:, i].mean():7.2f} us   p10 {np.percentile(seg[:, i], 10):7.2f}   p90 {np.percentile(seg[:, i], 90):7.2f}")
print(f"  entry times: p50 {np.percentile(us[:, 0], 50):.2f} us, p90 {np.percentile(us[:, 0], 90):.2f} us, last {us[:, 0].max():.2f} us")
