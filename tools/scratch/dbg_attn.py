import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd")); sys.path.insert(0, ROOT)
from sdm_hip import ops
torch.manual_seed(0)
for n, S, dk in ((4, 256, 512), (4, 64, 1024), (4, 256, 1024), (1, 256, 1024)):
    for mag in (1.0, 1e2, 1e4, 1e6, 1e8):
        qkv = (torch.randn(n * S, 3 * dk, device="cuda") * mag).contiguous()
        Sp = ops._pad4(S)
        res, P = ops.attention_core_small(qkv, n, S, 1, dk, Sp)
        ops.SMALL_ATTN[0] = False
        res2, P2 = ops.attention_core_fwd(qkv, n, S, 1, dk)
        ops.SMALL_ATTN[0] = True
        print(n, S, dk, mag, "finite", bool(torch.isfinite(res).all()), bool(torch.isfinite(res2).all()),
              "P nan", int(torch.isnan(P).sum()), int(torch.isnan(P2).sum()),
              "rel", float((res - res2).norm() / res2.norm()), "colsum", float(P.sum(1).min()), float(P.sum(1).max()))
