"""A/B of the AdaGN backward: SDM_GN_BWD_FUSED=0 (sums, coefficients, apply) against the default (sums, apply with the coefficients
rebuilt per workgroup) at the map sizes of the headline step (batch 4, fp32) and the C4 step (batch 32, bf16); replayed from a
hipGraph of 10 calls.      python tools/gn_bwd_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
    from sdm_hip import ops
    for dt, n, shapes in ((torch.float32, 4, [(64, 128), (32, 256), (16, 512), (8, 512), (8, 1024), (4, 1024)]),
                          (torch.bfloat16, 32, [(128, 128), (64, 256), (32, 512), (16, 512), (16, 1024), (8, 1024)])):
        for h, c in shapes:
            hw = h * h
            y = torch.randn(n * hw, c, device="cuda").to(dt)
            z = torch.randn(n * hw, c, device="cuda").to(dt)
            dout = torch.randn(n * hw, c, device="cuda").to(dt)
            gamma, beta = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
            mod = torch.randn(n, c, device="cuda")
            _, _, mean, rstd = ops.gn_affine(ops.Act(y, n, h, h), gamma, beta, mod, want_stats=True)
            dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
            f = lambda: ops.gn_adagn_bwd(dout, y, z, n, hw, gamma, beta, mod, mean, rstd, dg, db)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with ops.capture_scope(), torch.cuda.graph(g):
                for _ in range(10):
                    f()
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                a.record(); g.replay(); b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 10)
            print(f"{str(dt)[6:]:9s} n={n:2d} {h:3d}x{h:<3d} C={c:4d}: {best * 1e3:7.1f} us", flush=True)
    sys.exit(0)
for v in ("0", "1"):
    print(f"== SDM_GN_BWD_FUSED={v}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SDM_GN_BWD_FUSED=v), check=False)
