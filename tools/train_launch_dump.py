"""Per-launch dump of one training step (side stream off, HIP events per contraction launch):
    SDM_PROF_DUMP=out.csv python tools/train_launch_dump.py {c2|c4|c5} {f32|bf16}"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import lib, ops  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg, res, n, sr = {"c2": (STANDARD, 64, 4, None), "c4": (STANDARD, 128, 32, None),
                   "c5": (dict(STANDARD, in_channel=6, image_recon=True), 256, 8, (64, 256, 100))}[which]
torch.manual_seed(0)
net = U_Net(**cfg).cuda()
tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-5, seed=1, sr=sr,
                      train_dtype=torch.bfloat16 if dtype == "bf16" else None)
x0 = torch.randn(n, 3, res, res, device="cuda")
ops.SIDE.enabled = False
for _ in range(2):
    tr.step(x0)
torch.cuda.synchronize()
lib.prof_begin(100000)
tr.step(x0)
torch.cuda.synchronize()
p = lib.prof_end()
print({k: (round(v["ms"], 3), v["launches"], round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)) for k, v in p.items() if v["launches"]})
