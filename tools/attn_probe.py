"""One bf16 attention core (scores, column softmax, PV) repeatedly -- profiling target for the PMC passes:
    python tools/attn_probe.py n S d reps [twopass=1]       (default 8 4096 512 4)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops  # noqa: E402

n, S, d, reps = (list(map(int, sys.argv[1:5])) + [8, 4096, 512, 4][len(sys.argv) - 1:])[:4]
if len(sys.argv) > 5:
    ops.TWO_PASS_ATTN[0] = sys.argv[5] != "0"
qkv = (torch.randn(n * S, 3 * d, device="cuda") * 0.3).to(torch.bfloat16)
for _ in range(reps):
    ops.attention_core_fwd_bf16(qkv, n, S, 1, d)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    a.record()
    ops.attention_core_fwd_bf16(qkv, n, S, 1, d)
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
flops = 4.0 * n * S * S * d
print(f"attention core fwd bf16 n={n} S={S} d={d} two_pass={ops.TWO_PASS_ATTN[0]}: {best * 1e3:.1f} us "
      f"({flops / best / 1e9:.0f} algorithmic TFLOP/s)")
