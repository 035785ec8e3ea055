# what the vendor GEMM reaches on the contraction shapes of the bf16 step (plain GEMM, no im2col): a ceiling reference
import torch, time
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for M, N, K in ((32768, 512, 4608), (131072, 256, 2304), (524288, 128, 1152), (8192, 1024, 9216), (32768, 512, 512), (4608, 512, 32768), (1152, 128, 524288)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    dt = t(lambda: torch.matmul(a, b.t()))
    print(f"M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
    del a, b
