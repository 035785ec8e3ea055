"""A/B of sdm_wgrad_bf16 switches on one 3x3 conv's weight gradient: each variant (a set of environment variables, read once
per process by the library) runs in its own subprocess on the same GPU; dw / dbias are compared with the first variant and
with a torch fp32 reference on a small slice, then the launch is timed (hipGraph of 10 launches, replayed).

    python tools/wgrad_ab.py NxHxWxCINxCOUT [...] -- VAR=V[,VAR=V] [VAR=V ...]      ("-" = library defaults)
e.g. python tools/wgrad_ab.py 8x256x256x128x128 8x64x64x512x512 -- SDM_WGRAD8=0 -
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
    from models.custom_layers import TapConv, param_grad_like
    from sdm_hip import ops
    n, h, w, cin, cout = map(int, sys.argv[3:8])
    torch.manual_seed(0)
    conv = TapConv(cin, cout, "c3s1").cuda()
    x = ops.Act(torch.randn(n * h * w, cin, device="cuda").to(torch.bfloat16), n, h, w)
    g = torch.randn(n * h * w, cout, device="cuda").to(torch.bfloat16)
    gw = param_grad_like(conv.weight)
    gb = torch.zeros(cout, device="cuda")
    ops.conv_wgrad(g, (h, w), x, "c3s1", conv.grad_taps(gw), accumulate=1, dbias=gb)
    torch.cuda.synchronize()
    torch.save({"dw": gw.detach().float().cpu().contiguous(), "db": gb.cpu()}, sys.argv[2])

    def timed(f, reps=10):
        for _ in range(3): f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with ops.capture_scope() as kept, torch.cuda.graph(gr):
            for _ in range(reps): f()
        gr.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): gr.replay()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / (3 * reps) * 1e-3
    fl = 2.0 * n * h * w * cin * cout * 9
    t1 = timed(lambda: ops.conv_wgrad(g, (h, w), x, "c3s1", conv.grad_taps(gw), accumulate=1, dbias=gb))
    print(f"wgrad {t1*1e6:8.1f} us {fl/t1/1e12:7.1f} TFLOP/s", flush=True)
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
        f = f"/tmp/wgrad_ab_{os.getpid()}.pt"
        r = subprocess.run([sys.executable, __file__, "child", f] + shape.split("x"), env=env, capture_output=True, text=True, timeout=300)
        line = (r.stdout.strip().splitlines() or ["?"])[-1]
        if r.returncode != 0:
            print(shape, var, "FAILED", r.stderr[-800:], flush=True)
            continue
        o = torch.load(f)
        if ref is None:
            ref, d, db = o, 0.0, 0.0
        else:
            d = ((o["dw"] - ref["dw"]).norm() / ref["dw"].norm()).item()
            db = ((o["db"] - ref["db"]).norm() / ref["db"].norm()).item()
        print(f"{shape} [{var}]: {line}  rel dw vs first {d:.3g}  rel dbias {db:.3g}", flush=True)
