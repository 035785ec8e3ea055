"""Fixed cost of one fp32 igemm launch: tiny problems in a dependent chain (rocprofv3 --kernel-trace --stats gives the
per-kernel durations; the wall time / launch is printed here)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops
torch.manual_seed(0)
for (M, N, K) in ((64, 64, 64), (64, 64, 512), (64, 512, 512), (64, 512, 4608), (16, 16, 1024), (256, 512, 512), (4096, 128, 1152)):
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.zeros(N, device="cuda")
    y = ops.linear_rows(x, W, b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    reps = 50
    outs = [torch.empty(M, N, device="cuda") for _ in range(2)]
    with ops.capture_scope() as kept, torch.cuda.graph(g):
        for i in range(reps):
            ops.linear_rows(x, W, b, out=outs[i & 1])
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 / reps
    print(f"M={M} N={N} K={K}: {dt * 1e6:.2f} us per linear_rows call (graph of {reps})")
