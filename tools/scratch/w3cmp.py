# 3-tap weight-gradient kernel (SDM_WGRAD3=1) against the per-tap kernels (=0): dW, bias gradient; and timing
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
    from sdm_hip import ops
    n, h, w, cin, cout, c2, acc = map(int, sys.argv[3:10])
    g_ = torch.Generator().manual_seed(n + h + w + cin + cout)
    x = torch.randn(n * h * w, cin, generator=g_).cuda().to(torch.bfloat16)
    x2 = torch.randn(n * h * w, c2, generator=g_).cuda().to(torch.bfloat16) if c2 else None
    g = torch.randn(n * h * w, cout, generator=g_).cuda().to(torch.bfloat16)
    dw = torch.zeros(9, cout, cin + c2, device="cuda")
    if acc == 1:
        dw += 0.25
    db = torch.zeros(cout, device="cuda")
    xa = ops.Act(x, n, h, w, t2=x2)
    ops.conv_wgrad(g, (h, w), xa, "c3s1", dw, accumulate=acc, dbias=db)
    torch.cuda.synchronize()
    torch.save({"dw": dw.cpu(), "db": db.cpu()}, sys.argv[2])
    dw2 = torch.zeros_like(dw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): ops.conv_wgrad(g, (h, w), xa, "c3s1", dw2, accumulate=1)
    a.record()
    for _ in range(20): ops.conv_wgrad(g, (h, w), xa, "c3s1", dw2, accumulate=1)
    b.record(); torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 20 * 1e-3
    print(f"{dt*1e6:8.1f} us {2.0*n*h*w*(cin+c2)*cout*9/dt/1e12:7.1f} TFLOP/s", flush=True)
    sys.exit(0)
import torch
for shape in sys.argv[1:]:
    ref = None
    for mode in ("0", "1", ""):
        env = dict(os.environ)
        if mode: env["SDM_WGRAD3"] = mode
        else: env.pop("SDM_WGRAD3", None)
        f = f"/tmp/w3_{mode or 'd'}.pt"
        r = subprocess.run([sys.executable, __file__, "child", f] + shape.split("x"), env=env, capture_output=True, text=True, timeout=300)
        line = (r.stdout.strip().splitlines() or ["?"])[-1]
        if r.returncode != 0:
            print(shape, mode, "FAILED", r.stderr[-600:]); continue
        o = torch.load(f)
        if ref is None:
            ref = o; e1 = e2 = 0.0
        else:
            e1 = ((o["dw"] - ref["dw"]).norm() / ref["dw"].norm()).item()
            e2 = ((o["db"] - ref["db"]).norm() / ref["db"].norm()).item()
        print(f"{shape} WGRAD3={mode or 'plan'}: {line}  rel dW {e1:.2e}  rel db {e2:.2e}", flush=True)
