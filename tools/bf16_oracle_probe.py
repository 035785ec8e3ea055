"""Distance of the HIP bf16 paths to the oracle's bf16-storage mode (oracle/unet_oracle.py: Bf16Storage) and to the fp32
oracle, for the golden narrow nets (forward + gradients) and the standard architecture: the calibration behind the derived
gates of tests/test_gpu_bf16*.py.   python tools/bf16_oracle_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.U_Net import U_Net
from oracle import unet_oracle as O

GOLDEN = os.path.join(ROOT, "tests", "golden")
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def T(a):
    return torch.from_numpy(np.asarray(a))


for case in ("narrow_cond", "narrow_gain", "deep_uncond", "sr_recon", "no_emb"):
    g = np.load(os.path.join(GOLDEN, f"model_{case}.npz"))
    cfg = json.loads(str(g["cfg"]))
    P = O.seeded_params(cfg, seed=3, gain=float(g["gain"]))
    x, t = T(g["x"]), T(g["t"])
    cond = T(g["cond"]) if "cond" in g.files else None
    tt = t if cfg["time_dim"] is not None else None
    net = U_Net(**cfg)
    net.load_state_dict(P, strict=True)
    net = net.cuda().eval().set_inference_dtype(BF)
    with torch.no_grad():
        y = net(x.cuda(), None if tt is None else tt.cuda(), None if cond is None else cond.cuda())
        yb = O.unet_forward(P, cfg, x, tt, cond, storage=torch.bfloat16)
        ytw = O.unet_forward(P, cfg, x, tt, cond, storage=O.Bf16Storage().floor_twin())
        y32 = O.unet_forward(P, cfg, x, tt, cond)
    print(f"fwd  {case:12s}: hip vs bf16-oracle {rel(y, yb):.2e}   FLOOR (twin vs bf16-oracle) {rel(ytw, yb):.2e}   hip vs fp32 {rel(y, y32):.2e}   bf16-oracle vs fp32 {rel(yb, y32):.2e}", flush=True)
    # gradients
    tgt = T(g["target"])
    net.set_training_dtype(BF)
    loss, grads = net.loss_and_grads(x.cuda(), None if tt is None else tt.cuda(), None if cond is None else cond.cuda(), tgt.cuda())
    res = {}
    for nm, st in (("bf", O.Bf16Storage(train=True)), ("tw", O.Bf16Storage(train=True).floor_twin()), ("f32", None)):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        out = O.unet_forward(Pg, cfg, x, tt, cond, storage=st)
        l = torch.nn.functional.mse_loss(out, tgt)
        l.backward()
        res[nm] = (float(l), Pg)
    names = dict(net.named_parameters())
    eb, e32, ebo, etw = [], [], [], []
    for k, p in names.items():
        if p not in grads or res["f32"][1][k].grad is None:
            continue
        if ".conv_block_" in k and k.endswith("conv_layer.0.bias"):
            continue
        eb.append((rel(grads[p], res["bf"][1][k].grad), k))
        e32.append((rel(grads[p], res["f32"][1][k].grad), k))
        ebo.append((rel(res["bf"][1][k].grad, res["f32"][1][k].grad), k))
        etw.append((rel(res["tw"][1][k].grad, res["bf"][1][k].grad), k))
    ratio = sorted(a[0] / max(b[0], 1e-12) for a, b in zip(eb, etw))
    print(f"grad {case:12s}: per-tensor (hip vs bf16-oracle) / FLOOR: median {ratio[len(ratio)//2]:.2f}  p90 {ratio[int(len(ratio)*0.9)]:.2f}  worst {ratio[-1]:.2f}", flush=True)
    for lab, e in (("hip vs bf16-oracle", eb), ("FLOOR twin vs bf16-or.", etw), ("hip vs fp32", e32), ("bf16-oracle vs fp32", ebo)):
        e.sort()
        print(f"grad {case:12s}: {lab:20s} median {e[len(e)//2][0]:.2e}  p90 {e[int(len(e)*0.9)][0]:.2e}  worst {e[-1][0]:.2e} ({e[-1][1]})", flush=True)
    print(f"loss {case:12s}: hip {float(loss):.6f}  bf16-oracle {res['bf'][0]:.6f}  fp32 {res['f32'][0]:.6f}", flush=True)

cfg = dict(O.STANDARD_CFG)
torch.manual_seed(0)
net = U_Net(**cfg)
P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
net = net.cuda().eval().set_inference_dtype(BF)
gg = torch.Generator().manual_seed(1234)
x = torch.randn(2, 3, 64, 64, generator=gg)
t = torch.tensor([753, 12])
with torch.no_grad():
    y = net(x.cuda(), t.cuda())
    yb = O.unet_forward(P, cfg, x, t, storage=torch.bfloat16)
    ytw = O.unet_forward(P, cfg, x, t, storage=O.Bf16Storage().floor_twin())
    y32 = O.unet_forward(P, cfg, x, t)
print(f"fwd  standard 64x64 N=2: hip vs bf16-oracle {rel(y, yb):.2e}   FLOOR {rel(ytw, yb):.2e}   hip vs fp32 {rel(y, y32):.2e}   bf16-oracle vs fp32 {rel(yb, y32):.2e}", flush=True)
