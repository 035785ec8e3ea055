"""N-image DDPM sampling step of the standard U-Net, eager or hipGraph (profiling target).
    python tools/sample_probe.py [n=1] [steps=50] [f32|bf16] [graph|eager]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
import diffusion_sampling_algorithms as S  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dt = sys.argv[3] if len(sys.argv) > 3 else "f32"
mode = sys.argv[4] if len(sys.argv) > 4 else "graph"
torch.manual_seed(0)
net = U_Net(**STANDARD).cuda().eval()
if dt == "bf16":
    net.set_inference_dtype(torch.bfloat16)
sched = degraders.CosineNoiseDegradation(1000)
xT = torch.randn(n, 3, 64, 64, device="cuda")
quiet = lambda *a, **k: None
S.ddpm_sampling(net, sched, xT, 1, 4, device="cuda", log=quiet, use_graph=False)
torch.cuda.synchronize()
l0 = lib.launch_count()
S.ddpm_sampling(net, sched, xT, 1, 1, device="cuda", log=quiet, use_graph=False)
print("library launches per step:", lib.launch_count() - l0)
torch.cuda.synchronize()
t0 = time.perf_counter()
S.ddpm_sampling(net, sched, xT, 1, steps, device="cuda", log=quiet, use_graph=(mode == "graph"))
torch.cuda.synchronize()
print(f"n={n} {dt} {mode}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
