"""Per-launch dump (class, M, N, K, taps, split-K, us, TFLOP/s) of one bf16 forward of the standard U-Net at a given shape:
    SDM_PROF_DUMP=out.csv python tools/fwd_launch_dump_bf16.py [res=128] [n=128] [cond_dim=10] [forwards=1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import lib  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cond = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "0" else None
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.manual_seed(0)
net = U_Net(**dict(STANDARD, cond_dim=cond)).cuda().eval()
net.set_inference_dtype(torch.bfloat16)
x = torch.randn(n, 3, res, res, device="cuda")
t = torch.randint(1, 1000, (n,), device="cuda")
c = None if cond is None else torch.nn.functional.one_hot(torch.randint(0, cond, (n,)), cond).float().cuda()
with torch.no_grad():
    for _ in range(3):
        net(x, t, c)
    torch.cuda.synchronize()
    lib.prof_begin(30000)
    for _ in range(reps):                       # the CSV holds the forwards back to back: launch i of each is row i mod (rows / reps)
        net(x, t, c)
    torch.cuda.synchronize()
    p = lib.prof_end()
print({k: (round(v["ms"] * 1e3, 1), v["launches"]) for k, v in p.items()})
