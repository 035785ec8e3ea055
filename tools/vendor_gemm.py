"""Vendor bf16 GEMM (torch.matmul -> hipBLASLt) on the plain-GEMM equivalents of the conv shapes: the yardstick the bf16
contraction kernels are held against, and the target of the rocprofv3 passes behind profiles/r04_vendor_gemm_pmc.json.

    python tools/vendor_gemm.py [MxNxK ...]      (default: the 512-channel 3x3 conv on 8 x 64x64)
"""
import sys
import torch


def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


shapes = [tuple(map(int, s.split("x"))) for s in sys.argv[1:]] or [(32768, 512, 4608)]
for M, N, K in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    dt = t(lambda: torch.matmul(a, b.t()))
    print(f"M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
    del a, b
