# loss-goes-down soak of the bf16 training path on the standard architecture (fixed blocky batch, 64x64, batch 16):
# exercises the row-patch conv kernel, the three-tap weight gradients and the 256x256 tiles together over many steps
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd")); sys.path.insert(0, ROOT)
from bench import STANDARD
import degraders
from models.U_Net import U_Net
from sdm_hip.trainer import DiffusionTrainer
torch.manual_seed(0)
net = U_Net(**STANDARD).cuda()
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else None
tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=1e-4, seed=3, train_dtype=dt)
g = torch.Generator(device="cuda").manual_seed(1)
x0 = torch.rand(16, 3, 64, 64, device="cuda", generator=g) * 2 - 1
x0 = torch.nn.functional.avg_pool2d(x0, 8).repeat_interleave(8, 2).repeat_interleave(8, 3)
hist = []
for i in range(200):
    hist.append(float(tr.step(x0)))
    if i % 40 == 39:
        print(i + 1, sum(hist[-40:]) / 40, flush=True)
assert all(h == h for h in hist), "NaN"
print("first 20:", sum(hist[:20]) / 20, "last 20:", sum(hist[-20:]) / 20)
