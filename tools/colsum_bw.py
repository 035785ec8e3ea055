"""HBM rate of the column-sum kernels (bias gradients of the transposed convs) at the sizes the training steps run them:
    python tools/colsum_bw.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops  # noqa: E402

for dt, shapes in ((torch.float32, [(4 * 64 * 64, 128), (4 * 32 * 32, 256), (4 * 16 * 16, 512), (4 * 8 * 8, 512), (32 * 128 * 128, 128)]),
                   (torch.bfloat16, [(32 * 128 * 128, 128), (32 * 64 * 64, 256), (32 * 32 * 32, 512), (8 * 256 * 256, 128)])):
    for rows, c in shapes:
        x = torch.randn(rows, c, device="cuda").to(dt)
        out = torch.zeros(c, device="cuda")
        for _ in range(3):
            ops.colsum(x, out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.colsum(x, out)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print(f"{str(dt)[6:]:9s} {rows:7d} x {c:4d}: {us:7.1f} us  {x.numel() * x.element_size() / us / 1e6:5.2f} TB/s", flush=True)
