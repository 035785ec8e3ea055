"""A/B of sdm_igemm_bf16 switches on one 3x3 conv: each variant (a set of environment variables, read once per process by
the library) runs in its own subprocess on the same GPU; outputs (+ GroupNorm sums) are compared with the first variant,
then the forward conv and its data gradient are timed.

    python tools/tile_ab.py NxHxWxCINxCOUT [...] -- VAR=V[,VAR=V] [VAR=V ...]      ("-" = library defaults)
e.g. python tools/tile_ab.py 8x64x64x512x512 -- - SDM_BF16_8PH=0
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
    from models.custom_layers import RunCtx, TapConv
    from sdm_hip import ops
    n, h, w, cin, cout = map(int, sys.argv[3:8])
    torch.manual_seed(0)
    conv = TapConv(cin, cout, "c3s1").cuda()
    ctx = RunCtx(True, bf16=True)
    x = ops.Act(torch.randn(n * h * w, cin, device="cuda").to(torch.bfloat16), n, h, w)
    wt = conv.taps_for(ctx)
    y = ops.conv_fwd(x, wt, conv.bias, "c3s1", ops.ACT_SILU, gn_stats=True)
    torch.cuda.synchronize()
    out = {"y": y.t.float().cpu()}
    if y.stats is not None:
        out["stats"] = y.stats.clone().cpu()
        y.stats.zero_()
    torch.save(out, sys.argv[2])

    def timed(f, reps=10):
        """reps launches captured in a hipGraph and replayed: no host work between the kernels (a 3x3 conv call costs ~100 us
        of Python + ctypes, more than the faster kernels take)."""
        for _ in range(3): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with ops.capture_scope() as kept, torch.cuda.graph(g):
            for _ in range(reps): f()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): g.replay()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / (3 * reps) * 1e-3
    fl = 2.0 * n * h * w * cin * cout * 9
    t1 = timed(lambda: ops.conv_fwd(x, wt, conv.bias, "c3s1", ops.ACT_SILU))
    print(f"fwd {t1*1e6:8.1f} us {fl/t1/1e12:7.1f} TFLOP/s", flush=True)
    sys.exit(0)

import torch
args = sys.argv[1:]
cut = args.index("--")
shapes, variants = args[:cut], args[cut + 1:]
for shape in shapes:
    ref = None
    for var in variants:
        env = dict(os.environ)
        if var != "-":
            for kv in var.split(","):
                k, v = kv.split("=")
                env[k] = v
        f = f"/tmp/tile_ab_{os.getpid()}.pt"
        r = subprocess.run([sys.executable, __file__, "child", f] + shape.split("x"), env=env, capture_output=True, text=True, timeout=300)
        line = (r.stdout.strip().splitlines() or ["?"])[-1]
        if r.returncode != 0:
            print(shape, var, "FAILED", r.stderr[-600:], flush=True)
            continue
        o = torch.load(f)
        if ref is None:
            ref, d, ds = o, 0.0, 0.0
        else:
            d = (o["y"] - ref["y"]).abs().max().item()
            ds = ((o["stats"] - ref["stats"]).abs().max().item() / ref["stats"].abs().max().item()
                  if o.get("stats") is not None and ref.get("stats") is not None else -1)
        print(f"{shape} [{var}]: {line}  max|dy| vs first {d:.3g}  stats rel {ds:.3g}", flush=True)
