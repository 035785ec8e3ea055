"""Achieved HBM rate of the bf16 AdaGN application pass (sdm_affine_residual_bf16) at the C3 / C4 / C5 map sizes:
    python tools/affine_bw.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops  # noqa: E402

BF = torch.bfloat16
for n, hw, c in [(128, 128 * 128, 128), (128, 64 * 64, 256), (128, 32 * 32, 512), (128, 16 * 16, 1024), (32, 128 * 128, 128),
                 (8, 256 * 256, 128)]:
    h = int(hw ** 0.5)
    y = ops.Act(torch.randn(n * hw, c, device="cuda").to(BF), n, h, h)
    x = ops.Act(torch.randn(n * hw, c, device="cuda").to(BF), n, h, h)
    sc, sh = torch.randn(n, c, device="cuda"), torch.randn(n, c, device="cuda")
    for res in (None, x):
        for _ in range(3):
            o = ops.affine_residual(y, sc, sh, res)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            o = ops.affine_residual(y, sc, sh, res)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        by = n * hw * c * 2 * (3 if res is not None else 2)
        print(f"{n:4d} x {h}x{h} x {c:4d} residual={res is not None}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s", flush=True)
