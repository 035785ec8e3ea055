"""Per-launch contraction timing of eager forwards (tuning aid): python tools/profile_forward_launches.py batch res"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.U_Net import U_Net
from sdm_hip import lib
n, res = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
net = U_Net(num_resnet_blocks=1, num_layers=4, attn_layers=[2, 3], time_dim=512, min_channel=128, max_channel=512).cuda().eval()
x = torch.randn(n, 3, res, res, device="cuda"); t = torch.tensor([500], device="cuda")
with torch.no_grad():
    for _ in range(3): net(x, t)
    torch.cuda.synchronize()
    lib.prof_begin(100000)
    for _ in range(5): net(x, t)
    torch.cuda.synchronize()
    p = lib.prof_end()
print({k: (round(v["ms"] / 5, 3), v["launches"] // 5) for k, v in p.items()})
