"""A/B of the short-sequence attention cores against the generic path by batch size (which one should run at batch 64-128?)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for dt in (torch.float32, torch.bfloat16):
    for (S, d) in ((256, 512), (64, 512), (64, 1024), (256, 1024), (16, 1024)):
        for n in (1, 4, 16, 32, 128):
            qkv = (torch.randn(n * S, 3 * d, device="cuda") * 0.3).to(dt)
            f = ops.attention_core_fwd_bf16 if dt == torch.bfloat16 else (lambda q, n_, S_, h, d_, keep_p=True: ops.attention_core_fwd(q, n_, S_, h, d_, keep_p=keep_p))
            res = {}
            for name, small, keep in (("nograd", True, False), ("small+P", True, True), ("generic", False, True)):
                ops.SMALL_ATTN[0] = small
                res[name] = t(lambda: f(qkv, n, S, 1, d, keep_p=keep))
            ops.SMALL_ATTN[0] = True
            print(f"{str(dt)[6:]:9s} S={S:4d} d={d:5d} n={n:4d}: " + "  ".join(f"{k} {v:8.1f} us" for k, v in res.items()), flush=True)
