"""Forward (= one sampling step) of the standard U-Net at a given shape (profiling target):
    python tools/fwd_probe.py res batch {f32|bf16} [cond_dim] [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
from models.U_Net import U_Net  # noqa: E402

res, n = int(sys.argv[1]), int(sys.argv[2])
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
cond = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "0" else None
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
torch.manual_seed(0)
net = U_Net(**dict(STANDARD, cond_dim=cond)).cuda().eval()
if dt == "bf16":
    net.set_inference_dtype(torch.bfloat16)
x = torch.randn(n, 3, res, res, device="cuda")
t = torch.randint(1, 1000, (n,), device="cuda")
c = None if cond is None else torch.nn.functional.one_hot(torch.randint(0, cond, (n,)), cond).float().cuda()
with torch.no_grad():
    for _ in range(2):
        net(x, t, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        net(x, t, c)
    torch.cuda.synchronize()
print(f"forward {res}x{res} batch {n} {dt}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")
