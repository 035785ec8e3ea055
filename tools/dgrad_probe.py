"""Forward conv vs data-gradient conv of the same square layer (same M, N, K): the data gradient reads the weights
K-major (transposed access), the forward K-contiguous.  MI355X only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.custom_layers import TapConv  # noqa: E402
from sdm_hip import ops  # noqa: E402

SHAPES = [(4, 64, 64, 128), (4, 32, 32, 256), (4, 16, 16, 512), (4, 8, 8, 512), (4, 8, 8, 1024), (4, 4, 4, 1024),
          (4, 16, 16, 1024), (4, 32, 32, 512)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3


def main():
    for n, h, w, c in SHAPES:
        conv = TapConv(c, c, "c3s1").cuda()
        x = ops.Act(torch.randn(n * h * w, c, device="cuda"), n, h, w)
        g = torch.randn(n * h * w, c, device="cuda")
        tf = timeit(lambda: ops.conv_fwd(x, conv.taps(), conv.bias, "c3s1"))
        td = timeit(lambda: ops.conv_dgrad(g, (h, w), conv.taps(), "c3s1", (h, w), n))
        fl = 2.0 * n * h * w * c * c * 9
        print(f"c{c}@{h} batch {n}: fwd {tf:7.1f} us {fl/tf/1e6:6.1f} T | dgrad {td:7.1f} us {fl/td/1e6:6.1f} T | x{td/tf:.2f}", flush=True)


if __name__ == "__main__":
    main()
