"""Tile/split-K sweep of the contraction kernel on representative U-Net shapes (tuning aid; MI355X only):
    python tools/sweep_igemm.py [cfgs e.g. 1,5] [split-K factors e.g. 1,2,4]
Each configuration runs in a fresh subprocess because the override is read once per process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # name, n, h, w, cin, cout, kind
    ("c128@64", 4, 64, 64, 128, 128, "c3s1"), ("c256@32", 4, 32, 32, 256, 256, "c3s1"),
    ("c512@16", 4, 16, 16, 512, 512, "c3s1"), ("c512@8", 4, 8, 8, 512, 512, "c3s1"),
    ("c1024@4", 4, 4, 4, 1024, 1024, "c3s1"), ("c1024@8", 4, 8, 8, 1024, 1024, "c3s1"),
    ("c1024@16", 4, 16, 16, 1024, 1024, "c3s1"), ("c512@32", 4, 32, 32, 512, 512, "c3s1"),
    ("ct1024-256@16", 4, 16, 16, 1024, 256, "ct4"), ("c128@128b8", 8, 128, 128, 128, 128, "c3s1"),
]

CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from sdm_hip import ops
from models.custom_layers import TapConv
res = {}
for name, n, h, w, cin, cout, kind in %r:
    conv = TapConv(cin, cout, kind).cuda()
    x = ops.Act(torch.randn(n*h*w, cin, device="cuda"), n, h, w)
    for _ in range(3): y = ops.conv_fwd(x, conv.taps(), conv.bias, kind)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): y = ops.conv_fwd(x, conv.taps(), conv.bias, kind)
    b.record(); torch.cuda.synchronize()
    taps = 16 if kind == "ct4" else 9
    fl = 2.0*n*h*w*cin*cout*taps
    res[name] = [a.elapsed_time(b)/20*1e3, fl/(a.elapsed_time(b)/20*1e-3)/1e12]
print(json.dumps(res))
'''

N1_SHAPES = [  # the weight-streaming launches of the N=1 sampling step (python tools/sweep_igemm.py 1,5 4,8,16,32 n1)
    ("c512@16", 1, 16, 16, 512, 512, "c3s1"), ("c512@8", 1, 8, 8, 512, 512, "c3s1"), ("c1024@8", 1, 8, 8, 1024, 1024, "c3s1"),
    ("c1024@16", 1, 16, 16, 1024, 1024, "c3s1"), ("c512@32", 1, 32, 32, 512, 512, "c3s1"), ("c256@32", 1, 32, 32, 256, 256, "c3s1"),
    ("ct1024-512@8", 1, 8, 8, 1024, 512, "ct4"), ("ct1024-256@16", 1, 16, 16, 1024, 256, "ct4"),
]


def main():
    global SHAPES
    if len(sys.argv) > 3 and sys.argv[3] == "n1":
        SHAPES = N1_SHAPES
    out = {}
    cfgs = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 3, 5]     # 5 = 64x128
    sks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
    for cfg in cfgs:
        for sk in sks:
            env = dict(os.environ, SDM_IGEMM_FORCE=f"{cfg},{sk}")
            r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)],
                               env=env, capture_output=True, text=True)
            try:
                out[f"{cfg},{sk}"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                out[f"{cfg},{sk}"] = {"error": r.stderr[-400:]}
    names = [s[0] for s in SHAPES]
    print("cfg,sk  " + "  ".join(f"{n:>14s}" for n in names))
    for k, v in out.items():
        if "error" in v:
            print(k, v["error"]); continue
        print(f"{k:7s} " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names))
    r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)], capture_output=True, text=True)
    v = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"{'auto':7s} " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names))

if __name__ == "__main__":
    main()
