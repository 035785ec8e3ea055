# compare the forced tile configurations of sdm_igemm_bf16 on one conv (outputs + GroupNorm sums) and time them
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    y = ops.conv_fwd(x, wt, conv.bias, "c3s1", gn_stats=True)
    torch.cuda.synchronize()
    out = {"y": y.t.float().cpu()}
    if y.stats is not None:
        out["stats"] = y.stats.clone().cpu() if hasattr(y.stats, "clone") else None
    torch.save(out, sys.argv[2])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): ops.conv_fwd(x, wt, conv.bias, "c3s1")
    a.record()
    for _ in range(20): ops.conv_fwd(x, wt, conv.bias, "c3s1")
    b.record(); torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 20 * 1e-3
    print(f"{dt*1e6:8.1f} us {2.0*n*h*w*cin*cout*9/dt/1e12:7.1f} TFLOP/s", flush=True)
    sys.exit(0)
import torch
for shape in sys.argv[1:]:
    ref = None
    for force in ("0", "d"):
        env = dict(os.environ)
        if force != "d": env["SDM_BF16_ROW"] = force
        f = f"/tmp/tile_{force[0]}.pt"
        r = subprocess.run([sys.executable, __file__, "child", f] + shape.split("x"), env=env, capture_output=True, text=True, timeout=200)
        line = (r.stdout.strip().splitlines() or ["?"])[-1]
        if r.returncode != 0:
            print(shape, force, "FAILED", r.stderr[-400:]); continue
        o = torch.load(f)
        if ref is None:
            ref = o; d = 0.0; ds = 0.0
        else:
            d = (o["y"] - ref["y"]).abs().max().item()
            ds = (o["stats"] - ref["stats"]).abs().max().item() / ref["stats"].abs().max().item() if o.get("stats") is not None and ref.get("stats") is not None else -1
        print(f"{shape} cfg {force}: {line}  max|dy| vs cfg0 {d:.3g}  stats rel {ds:.3g}", flush=True)
