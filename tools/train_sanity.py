"""Loss-goes-down sanity of the fused trainer on a fixed synthetic batch (MI355X only): 200 steps, narrow U-Net."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
import degraders  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402


def main():
    torch.manual_seed(0)
    net = U_Net(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=64, cond_dim=None, num_layers=3,
                attn_layers=[2], num_heads=1, dim_per_head=None, min_channel=32, max_channel=128).cuda()
    tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-4, seed=3)
    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.rand(8, 3, 32, 32, device="cuda", generator=g) * 2 - 1
    x0 = torch.nn.functional.avg_pool2d(x0, 4).repeat_interleave(4, 2).repeat_interleave(4, 3)   # blocky images
    hist = []
    for i in range(300):
        hist.append(float(tr.step(x0)))
        if i % 50 == 49:
            print(f"step {i+1:4d}  mean loss of last 50: {sum(hist[-50:])/50:.4f}", flush=True)
    first, last = sum(hist[:30]) / 30, sum(hist[-30:]) / 30
    print(f"first-30 mean {first:.4f}  last-30 mean {last:.4f}")
    assert last < 0.5 * first, "loss did not fall"
    print("ok")


if __name__ == "__main__":
    main()
