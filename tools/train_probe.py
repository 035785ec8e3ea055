"""One training configuration for a few steps (profiling target: rocprofv3 --kernel-trace --stats -- python tools/train_probe.py c5 bf16).
  python tools/train_probe.py {c2|c4|c5} {f32|bf16} [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import lib  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg, res, n, sr = {"c2": (STANDARD, 64, 4, None), "c4": (STANDARD, 128, 32, None),
                   "c5": (dict(STANDARD, in_channel=6, image_recon=True), 256, 8, (64, 256, 100))}[which]
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = U_Net(**cfg).to(dev)
tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-5, seed=1, sr=sr,
                      train_dtype=torch.bfloat16 if dtype == "bf16" else None)
x0 = torch.randn(n, 3, res, res, device=dev)
for _ in range(2):
    tr.step(x0)
torch.cuda.synchronize()
l0 = lib.launch_count()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step(x0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{which} {dtype}: {dt*1e3:.2f} ms/step, loss {float(loss):.5f}, {(lib.launch_count()-l0)//steps} launches/step, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
