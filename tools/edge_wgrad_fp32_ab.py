"""A/B of the fp32 edge-conv weight gradients (sdm_wgrad's thin-operand kernel against the matrix-core path, SDM_WGRAD_THIN=0), alone
on the chip, replayed from a hipGraph:      python tools/edge_wgrad_fp32_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
    from sdm_hip import ops
    for n, h, C in ((4, 64, 128), (32, 128, 128), (8, 256, 128)):
        for cin, cout in ((3, C), (C, 3)):
            xr = torch.randn(n * h * h, 4 if cin == 3 else cin, device="cuda")
            gr = torch.randn(n * h * h, 4 if cout == 3 else cout, device="cuda")
            dw = torch.zeros(9, cout, cin, device="cuda")
            db = torch.zeros(cout, device="cuda")
            f = lambda: ops.conv_wgrad(gr[:, :cout], (h, h), ops.Act(xr[:, :cin], n, h, h), "c3s1", dw, accumulate=1, dbias=db)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with ops.capture_scope(), torch.cuda.graph(g):
                for _ in range(10):
                    f()
            g.replay()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                a.record(); g.replay(); e.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(e) / 10)
            print(f"n={n:2d} {h}x{h} {cin:3d}->{cout:3d}: {best * 1e3:7.1f} us", flush=True)
    sys.exit(0)
for v in ("0", "1"):
    print(f"== SDM_WGRAD_THIN={v}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SDM_WGRAD_THIN=v), check=False)
