import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd")); sys.path.insert(0, ROOT)
from bench import STANDARD
import degraders as D, diffusion_sampling_algorithms as S
from models.U_Net import U_Net
from sdm_hip import ops
torch.manual_seed(0)
net = U_Net(**dict(STANDARD, cond_dim=10)).cuda().eval()
g = torch.Generator().manual_seed(77)
xT = torch.randn(2, 3, 128, 128, generator=g).cuda()
labels = torch.eye(10)[[4, 9]].cuda()
quiet = lambda *a, **k: None
res = {}
for small in (False, True):
    for table in (False, True):
        ops.SMALL_ATTN[0] = small; S.MOD_TABLE[0] = table
        out = S.ddim_sampling(net, D.CosineNoiseDegradation(1000), xT, 1, 1000, None, labels, 400, "cuda", quiet, use_graph=False, cfg_scale=3.0)
        res[(small, table)] = out
        print(small, table, bool(torch.isfinite(out).all()), float(out.abs().max()))
ref = res[(False, False)]
for k, v in res.items():
    print(k, float((v - ref).norm() / ref.norm()))
# single forward
ops.SMALL_ATTN[0] = True
with torch.no_grad():
    for n in (1, 2, 4):
        x = torch.randn(n, 3, 128, 128, device="cuda")
        t = torch.full((n,), 500, device="cuda")
        lab = torch.eye(10)[:n].cuda()
        ops.SMALL_ATTN[0] = False
        a = net(x, t, lab)
        ops.SMALL_ATTN[0] = True
        b = net(x, t, lab)
        print(n, bool(torch.isfinite(b).all()), float((a - b).norm() / a.norm()))
