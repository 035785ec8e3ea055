"""Run one bf16 conv shape repeatedly (profiling aid):
    python tools/one_conv_bf16.py n h w cin cout kind reps [fwd|dgrad|wgrad]
fwd: sdm_igemm_bf16 (forward); dgrad: sdm_igemm_bf16 against the transposed weight copy; wgrad: sdm_wgrad_bf16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.custom_layers import RunCtx, TapConv, param_grad_like  # noqa: E402
from sdm_hip import ops  # noqa: E402

n, h, w, cin, cout = map(int, sys.argv[1:6])
kind = sys.argv[6]
reps = int(sys.argv[7])
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
conv = TapConv(cin, cout, kind).cuda()
ctx = RunCtx(True, bf16=True)
x = ops.Act(torch.randn(n * h * w, cin, device="cuda").to(torch.bfloat16), n, h, w)
ho, wo = ops.conv_out_hw(h, w, kind)
g = torch.randn(n * ho * wo, cout, device="cuda").to(torch.bfloat16)
gw = param_grad_like(conv.weight)
taps = 16 if kind == "ct4" else 9
flops = 2.0 * n * ho * wo * cin * cout * 9 if kind != "ct4" else 2.0 * n * h * w * cin * cout * 16


def once():
    if mode == "fwd":
        ops.conv_fwd(x, conv.taps_for(ctx), conv.bias, kind)
    elif mode == "dgrad":
        ops.conv_dgrad(g, (ho, wo), conv.taps_t_for(ctx), kind, (h, w), n)
    else:
        ops.conv_wgrad(g, (ho, wo), x, kind, conv.grad_taps(gw), accumulate=1)


for _ in range(2):
    once()
torch.cuda.synchronize()
if os.environ.get("ONE_CONV_COLD") == "1":
    # every call behind 1 GiB of unrelated writes: weights and activations come from HBM, not from L2 / the Infinity Cache
    junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    # ONE_CONV_TLB=<GiB>: additionally touch one word per 4 KiB page of a buffer that large (address-translation caches cold too)
    tlb_gib = int(os.environ.get("ONE_CONV_TLB", "0"))
    pages = torch.empty(tlb_gib << 28, dtype=torch.float32, device="cuda") if tlb_gib else None
    tot = 0.0
    for i in range(reps):
        junk.fill_(float(i))
        if pages is not None:
            pages[::1024].fill_(float(i))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        once()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    us = tot / reps * 1e3
    print(f"{mode} {kind} n{n} {h}x{w} {cin}->{cout} COLD: {us:.1f} us per call (event pair around each), {flops / us / 1e6:.0f} TFLOP/s")
    sys.exit(0)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    once()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / reps * 1e3
print(f"{mode} {kind} n{n} {h}x{w} {cin}->{cout}: {us:.1f} us per call, {flops / us / 1e6:.0f} TFLOP/s")
