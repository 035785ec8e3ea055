"""Per-launch dump (class, M, N, K, taps, split-K, us, TFLOP/s) of one fp32 forward of the standard U-Net at batch n:
    SDM_PROF_DUMP=out.csv python tools/fwd_launch_dump.py [n=1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
net = U_Net(**STANDARD).cuda().eval()
x = torch.randn(n, 3, 64, 64, device="cuda")
t = torch.full((1,), 500, device="cuda")
with torch.no_grad():
    for _ in range(3):
        net(x, t)
    torch.cuda.synchronize()
    lib.prof_begin(10000)
    net(x, t)
    torch.cuda.synchronize()
    p = lib.prof_end()
print({k: (round(v["ms"] * 1e3, 1), v["launches"]) for k, v in p.items()})
