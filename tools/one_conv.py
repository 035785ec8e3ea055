"""One fp32 conv shape, repeated (profiling target): python tools/one_conv.py n h w cin cout kind [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.custom_layers import TapConv  # noqa: E402
from sdm_hip import ops  # noqa: E402

n, h, w, cin, cout = (int(v) for v in sys.argv[1:6])
kind = sys.argv[6] if len(sys.argv) > 6 else "c3s1"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
conv = TapConv(cin, cout, kind).cuda()
x = ops.Act(torch.randn(n * h * w, cin, device="cuda"), n, h, w)
for _ in range(3):
    y = ops.conv_fwd(x, conv.taps(), conv.bias, kind)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    y = ops.conv_fwd(x, conv.taps(), conv.bias, kind)
b.record()
torch.cuda.synchronize()
print(f"{kind} n{n} {h}x{w} {cin}->{cout}: {a.elapsed_time(b) / reps * 1e3:.1f} us per call")
