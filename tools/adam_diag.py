"""Diagnostic: fused trainer step vs torch.optim.Adam on the GPU path's own gradients, per tensor (standard U-Net)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd")); sys.path.insert(0, ROOT)
from oracle import unet_oracle as O
import degraders as D
from models.U_Net import U_Net
from sdm_hip.trainer import DiffusionTrainer
cfg = dict(O.STANDARD_CFG)
torch.manual_seed(0)
net = U_Net(**cfg)
P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
net = net.cuda()
g = torch.Generator().manual_seed(99)
x0 = torch.randn(2, 3, 64, 64, generator=g); eps = torch.randn(2, 3, 64, 64, generator=g); t = torch.tensor([812, 37])
sched = D.CosineNoiseDegradation(1000)
x_t = sched(x0.cuda(), t.cuda(), eps.cuda())
loss, grads = net.loss_and_grads(x_t, t.cuda(), None, eps.cuda())
torch.cuda.synchronize()
dead = set(O.dead_param_names(cfg))
names = [k for k, _ in net.named_parameters() if k not in dead]
byname = dict(net.named_parameters())
gcpu = {k: grads[byname[k]].detach().cpu().contiguous().clone() for k in names}
leaves = [P[k].clone().requires_grad_(True) for k in names]
opt = torch.optim.Adam(leaves, lr=1e-3, betas=(0.5, 0.999))
for leaf, k in zip(leaves, names): leaf.grad = gcpu[k]
opt.step()
tr = DiffusionTrainer(net, sched, lr=1e-3)
tr.step(x0.cuda(), eps=eps.cuda(), t=t.cuda())
torch.cuda.synchronize()
sd = net.state_dict()
rows = []
for leaf, k in zip(leaves, names):
    d = (sd[k].detach().cpu() - leaf.detach()).abs()
    n_off = int((d > 1e-4).sum())
    if n_off:
        idx = int(d.flatten().argmax())
        rows.append((n_off / d.numel(), k, d.numel(), float(d.max()), float(gcpu[k].flatten()[idx]), float(gcpu[k].abs().median())))
rows.sort(reverse=True)
for r in rows[:25]: print("%.4f %-60s n=%d maxdiff=%.2e g_at_max=%.3e median|g|=%.2e" % r)
print("tensors with diffs:", len(rows), "of", len(names))
