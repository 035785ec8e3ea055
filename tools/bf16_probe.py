"""bf16 sampling path timings (MI355X only): conv shapes under forced tile choices, and the forward (= one sampling
step) of the standard U-Net in fp32 vs bf16 storage.  python tools/bf16_probe.py [conv] [fwd]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "simple-diffusion-model_amd")
SHAPES = [  # name, n, h, w, cin, cout, kind
    ("c128@128b64", 64, 128, 128, 128, 128, "c3s1"), ("c256@64b64", 64, 64, 64, 256, 256, "c3s1"),
    ("c512@32b64", 64, 32, 32, 512, 512, "c3s1"), ("c1024@16b64", 64, 16, 16, 1024, 1024, "c3s1"),
    ("c128@64b4", 4, 64, 64, 128, 128, "c3s1"), ("c512@16b4", 4, 16, 16, 512, 512, "c3s1"),
    ("c1024@8b4", 4, 8, 8, 1024, 1024, "c3s1"), ("c512@4b4", 4, 4, 4, 512, 512, "c3s1"),
    ("ct1024-256@16b4", 4, 16, 16, 1024, 256, "ct4"),
]
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from sdm_hip import ops
from models.custom_layers import TapConv, RunCtx
res = {}
ctx = RunCtx(False, bf16=True)
for name, n, h, w, cin, cout, kind in %r:
    conv = TapConv(cin, cout, kind).cuda()
    x = ops.Act(torch.randn(n*h*w, cin, device="cuda").to(torch.bfloat16), n, h, w)
    wt = conv.taps_for(ctx)
    for _ in range(3): y = ops.conv_fwd(x, wt, conv.bias, kind)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): y = ops.conv_fwd(x, wt, conv.bias, kind)
    b.record(); torch.cuda.synchronize()
    taps = 16 if kind == "ct4" else 9
    fl = 2.0*n*h*w*cin*cout*taps
    res[name] = [a.elapsed_time(b)/10*1e3, fl/(a.elapsed_time(b)/10*1e-3)/1e12]
    del x, y
print(json.dumps(res))
'''


def conv():
    names = [s[0] for s in SHAPES]
    print("force  vepi " + "  ".join(f"{n:>15s}" for n in names))
    for force in ("0,1", "1,1", "auto"):
        for nb in ("0", "1"):
            env = dict(os.environ, SDM_BF16_VEC_EPI=nb)
            if force != "auto":
                env["SDM_BF16_FORCE"] = force
            r = subprocess.run([sys.executable, "-c", CHILD % (PKG, SHAPES)], env=env, capture_output=True, text=True)
            try:
                v = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"{force:6s} {nb:4s} " + "  ".join(f"{v[n][0]:8.1f}us{v[n][1]:5.0f}T" for n in names), flush=True)
            except Exception:
                print(force, nb, r.stderr[-600:], flush=True)


def fwd():
    import torch
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    from bench import STANDARD
    from models.U_Net import U_Net
    for res, n, cond in ((64, 1, None), (64, 4, None), (128, 64, 10), (128, 128, 10)):
        torch.manual_seed(0)
        net = U_Net(**dict(STANDARD, cond_dim=cond)).cuda().eval()
        x = torch.randn(n, 3, res, res, device="cuda")
        t = torch.randint(1, 1000, (n,), device="cuda")
        c = None if cond is None else torch.nn.functional.one_hot(torch.randint(0, cond, (n,)), cond).float().cuda()
        out = {}
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            net.set_inference_dtype(dt)
            with torch.no_grad():
                for _ in range(2):
                    net(x, t, c)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    net(x, t, c)
                torch.cuda.synchronize()
                out[name] = (time.perf_counter() - t0) / reps
        print(f"forward {res}x{res} batch {n}: fp32 {out['fp32']*1e3:8.2f} ms   bf16 {out['bf16']*1e3:8.2f} ms   "
              f"x{out['fp32']/out['bf16']:.2f}", flush=True)
        del net
        torch.cuda.empty_cache()


if __name__ == "__main__":
    what = sys.argv[1:] or ["conv", "fwd"]
    if "conv" in what:
        conv()
    if "fwd" in what:
        fwd()
