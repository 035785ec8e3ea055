"""Timing of the bf16 path's edge convolutions alone (stem: fp32 NCHW image -> bf16 rows; head: bf16 rows -> fp32 NCHW):
    python tools/edge_probe.py [stem|head|both] [n=128] [res=128] [channels=128]
Prints us per launch and the HBM rate of the algorithmic bytes (image + rows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models import custom_layers as L  # noqa: E402
from sdm_hip import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
res = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ch = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


rows_bytes = n * res * res * ch * 2
img_bytes = n * res * res * 3 * 4
if what in ("stem", "both"):
    conv = L.TapConv(3, ch, "c3s1").to(dev)
    x = torch.randn(n, 3, res, res, device=dev)
    us = timed(lambda: ops.stem_conv(x, conv.taps(), conv.bias, ops.ACT_SILU, out_dtype=torch.bfloat16))
    print(f"stem  {n}x3x{res}x{res} -> {ch} ch  SDM_STEM_NB={os.environ.get('SDM_STEM_NB', '-')}: {us:8.1f} us  "
          f"{(rows_bytes + img_bytes) / us / 1e6:6.2f} TB/s of algorithmic bytes")
if what in ("head", "both"):
    conv = L.TapConv(ch, 3, "c3s1").to(dev)
    ctx = L.RunCtx(False, bf16=True)
    xr = torch.randn(n * res * res, ch, device=dev).to(torch.bfloat16)
    w16 = conv.taps_for(ctx)
    act = ops.Act(xr, n, res, res)
    us = timed(lambda: ops.head_conv(act, conv.taps(), conv.bias, ops.ACT_NONE, wt16=w16))
    print(f"head  {n}x{ch}x{res}x{res} -> 3 ch: {us:8.1f} us  {(rows_bytes + img_bytes) / us / 1e6:6.2f} TB/s of algorithmic bytes")
