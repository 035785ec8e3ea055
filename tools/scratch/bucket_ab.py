"""A/B: gradient bucket size of the trainer (Adam granularity at world 1) on the headline step.  python tools/scratch/bucket_ab.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402

dev = torch.device("cuda:0")
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
net = U_Net(**STANDARD).to(dev)
tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-5, seed=1, bucket_mb=mb)
x0 = torch.randn(4, 3, 64, 64, device=dev)
for _ in range(3):
    tr.step(x0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    tr.step(x0)
torch.cuda.synchronize()
print(f"bucket_mb {mb}: {len(tr.reducer.buckets)} buckets, {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step")
