"""Eager forward passes of the standard U-Net (profiling aid): python tools/profile_forward.py batch res reps"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.U_Net import U_Net
n, res, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
net = U_Net(num_resnet_blocks=1, num_layers=4, attn_layers=[2, 3], time_dim=512, min_channel=128, max_channel=512).cuda().eval()
x = torch.randn(n, 3, res, res, device="cuda"); t = torch.tensor([500], device="cuda")
with torch.no_grad():
    for _ in range(3): y = net(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): y = net(x, t)
    torch.cuda.synchronize()
print(f"eager forward batch {n} res {res}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms")
