"""A/B of the bf16 contraction tile on SHORT-K problems (K = 512 / 1024 token linears and the attention core): the planner's
256x256-vs-128x128 rule was fitted on 3x3 convs (72+ K-tiles per output tile); at 8-16 K-tiles the fixed cost of a 256x256 tile
(one workgroup per CU: prologue, epilogue and hand-over are exposed) weighs more.  Each variant runs in its own subprocess
(the library reads its switches once).

    python tools/shortk_tile_ab.py            -> table on stdout
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LINEARS = [(131072, 512, 1536), (131072, 512, 512), (32768, 1024, 3072), (32768, 1024, 1024), (32768, 512, 1536),
           (32768, 512, 512), (8192, 1024, 3072), (8192, 1024, 1024)]
ATTN = [(8, 4096, 512), (128, 1024, 512), (32, 1024, 512), (128, 256, 1024)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
    from sdm_hip import ops

    def timed(f, reps=8):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with ops.capture_scope(), torch.cuda.graph(g):
            for _ in range(reps):
                f()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(4):
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps)
        return best * 1e3

    torch.manual_seed(0)
    for rows, K, O in LINEARS:
        x = (torch.randn(rows, K, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(O, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(O, device="cuda")
        out = torch.empty(rows, O, device="cuda", dtype=torch.bfloat16)
        us = timed(lambda: ops.linear_rows(x, W, b, out=out))
        print(f"linear {rows}x{K}->{O}: {us:8.1f} us {2.0 * rows * K * O / us / 1e6:7.0f} TFLOP/s  sum {float(out.float().sum()):.6g}", flush=True)
    for n, S, d in ATTN:
        qkv = (torch.randn(n * S, 3 * d, device="cuda") * 0.3).to(torch.bfloat16)
        res = ops.attention_core_fwd_bf16(qkv, n, S, 1, d)
        us = timed(lambda: ops.attention_core_fwd_bf16(qkv, n, S, 1, d), reps=3)
        print(f"attn n={n} S={S} d={d}: {us:8.1f} us {4.0 * n * S * S * d / us / 1e6:7.0f} TFLOP/s  sum {float(res.float().sum()):.6g}", flush=True)
    sys.exit(0)

variants = sys.argv[1:] or ["-", "SDM_BF16_FORCE=0;1", "SDM_BF16_8PH=0"]      # ";" stands for the "," inside a value
for v in variants:
    env = dict(os.environ)
    if v != "-":
        for kv in v.split(","):
            if "=" in kv:
                k, val = kv.split("=", 1)
                env[k] = val.replace(";", ",")
    print(f"== {v}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
