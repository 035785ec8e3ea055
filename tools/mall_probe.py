"""Does a weight-streaming conv of the N=1 sampling step run faster when its weights already sit in the Infinity Cache?
Times one conv (default 1 x 8x8, 1024 -> 1024, 3x3: 37.7 MB of fp32 weights, M = 64) back to back (weights hot in the 256 MB
memory-side cache) and with 512 MB of unrelated traffic in front of every call (weights cold: what a 489 MB-per-step fp32
sampling run sees), in fp32 and bf16.    python tools/mall_probe.py [n h w cin cout]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.custom_layers import RunCtx, TapConv  # noqa: E402
from sdm_hip import ops  # noqa: E402

n, h, w, cin, cout = (list(map(int, sys.argv[1:6])) + [1, 8, 8, 1024, 1024][len(sys.argv) - 1:])[:5]
torch.manual_seed(0)
conv = TapConv(cin, cout, "c3s1").cuda()
evict = torch.empty(128 << 20, dtype=torch.float32, device="cuda")      # 512 MB
evict2 = torch.empty_like(evict)
for bf16 in (False, True):
    ctx = RunCtx(False, bf16=bf16)
    x = torch.randn(n * h * w, cin, device="cuda")
    a = ops.Act(x.to(torch.bfloat16) if bf16 else x, n, h, w)
    wt = conv.taps_for(ctx)
    for _ in range(3):
        ops.conv_fwd(a, wt, conv.bias, "c3s1", ops.ACT_SILU)
    torch.cuda.synchronize()

    def timed(cold, reps=20):
        ts = []
        for _ in range(reps):
            if cold:
                evict2.copy_(evict)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv_fwd(a, wt, conv.bias, "c3s1", ops.ACT_SILU)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    hot, cold = timed(False), timed(True)
    mb = wt.numel() * wt.element_size() / 1e6
    print(f"{'bf16' if bf16 else 'fp32'} {n}x{h}x{w} {cin}->{cout}: weights {mb:.1f} MB; hot {hot:.1f} us ({mb / hot * 1e-3 * 1e3:.0f} GB/s)  "
          f"cold {cold:.1f} us ({mb / cold:.0f} GB/s)   [event pair included: ~6 us]", flush=True)
