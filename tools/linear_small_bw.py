"""Latency of the small-batch linears (ConditionalEmbedding MLP layers, AdaGN modulations) by batch:
    python tools/linear_small_bw.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops  # noqa: E402

for M in (1, 4, 8, 32, 128, 256):
    for K, O in ((512, 512), (512, 1024)):
        x = torch.randn(M, K, device="cuda")
        W, b = torch.randn(O, K, device="cuda") * K ** -0.5, torch.randn(O, device="cuda")
        f = lambda: ops.linear_small(x, W, b, act=ops.ACT_SILU, save_z=True)       # training form: the one-wave-per-feature kernel
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with ops.capture_scope(), torch.cuda.graph(g):
            for _ in range(10):
                f()
        g.replay()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            a.record(); g.replay(); e.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) / 10)
        ref = torch.nn.functional.silu(torch.nn.functional.linear(x, W, b))
        res = f()
        err = float(((res[0] if isinstance(res, tuple) else res) - ref).abs().max())
        print(f"M={M:3d} K={K} O={O:4d}: {best * 1e3:6.1f} us   max|d| vs torch {err:.2e}", flush=True)
