"""Timings of the other BASELINE.json configurations on one MI355X, fp32 (the bench line is configs[1]):
  * attention core (QK^T, column softmax, PV and its backward) at the 256^2 SR model's token counts -> MFMA utilisation
  * C3-shaped sampling step: 128^2, cond_dim=10, classifier-free guidance on (two U-Net passes), batch 64
  * C4-shaped per-GPU training step: 128^2, batch 32
  * C5-shaped per-GPU SR training step: 64->256, in_channel 6, tanh head, batch 8
Synthetic inputs, random-init weights.  python tools/config_probe.py [attn] [c3] [c4] [c5]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD, PEAK_FP32_MFMA_TFLOPS  # noqa: E402
import degraders  # noqa: E402
import diffusion_sampling_algorithms as S  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import ops  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def attn():
    for n, S_, C in ((8, 4096, 512), (8, 4096, 1024), (8, 1024, 512), (4, 1024, 512), (4, 256, 512)):
        qkv = torch.randn(n * S_, 3 * C, device=DEV)
        dres = torch.randn(n * S_, C, device=DEV)
        res, P = ops.attention_core_fwd(qkv, n, S_, 1, C)
        tf = timeit(lambda: ops.attention_core_fwd(qkv, n, S_, 1, C))
        tb = timeit(lambda: ops.attention_core_bwd(dres, qkv, P, n, S_, 1, C))
        ff = 2 * 2.0 * n * S_ * S_ * C            # QK^T + PV
        fb = 4 * 2.0 * n * S_ * S_ * C            # dV, dP, dQ, dK
        print(f"attention core n={n} tokens={S_} d={C}: fwd {tf*1e3:7.3f} ms {ff/tf/1e12:6.1f} TFLOP/s "
              f"({ff/tf/1e12/PEAK_FP32_MFMA_TFLOPS:4.0%} of fp32 MFMA peak) | bwd {tb*1e3:7.3f} ms {fb/tb/1e12:6.1f} TFLOP/s "
              f"({fb/tb/1e12/PEAK_FP32_MFMA_TFLOPS:4.0%})", flush=True)
        # the same core on the bf16 path (scores fp32, probabilities bf16, fp32 accumulation), against the 2.5 PFLOP/s roof
        q16, d16 = qkv.to(torch.bfloat16), dres.to(torch.bfloat16)
        _, P16 = ops.attention_core_fwd_bf16(q16, n, S_, 1, C, keep_p=True)
        tf16 = timeit(lambda: ops.attention_core_fwd_bf16(q16, n, S_, 1, C))
        tb16 = timeit(lambda: ops.attention_core_bwd_bf16(d16, q16, P16, n, S_, 1, C))
        print(f"   bf16: fwd {tf16*1e3:7.3f} ms {ff/tf16/1e12:6.1f} TFLOP/s ({ff/tf16/1e12/2500:5.1%} of bf16 MFMA peak) | "
              f"bwd {tb16*1e3:7.3f} ms {fb/tb16/1e12:6.1f} TFLOP/s ({fb/tb16/1e12/2500:5.1%})", flush=True)
        del qkv, dres, res, P, q16, d16, P16
        torch.cuda.empty_cache()


def c3():
    cfg = dict(STANDARD, cond_dim=10)
    torch.manual_seed(0)
    net = U_Net(**cfg).to(DEV).eval()
    sched = degraders.CosineNoiseDegradation(1000)
    n = 64
    noise = torch.randn(n, 3, 128, 128, device=DEV)
    cond = torch.nn.functional.one_hot(torch.randint(0, 10, (n,)), 10).float().to(DEV)
    steps = 10
    t0 = time.perf_counter()
    quiet = lambda *a, **k: None
    S.ddim_sampling(net, sched, noise.clone(), 1, 1000, None, cond, 1000 // steps, DEV, quiet, cfg_scale=2.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    S.ddim_sampling(net, sched, noise.clone(), 1, 1000, None, cond, 1000 // steps, DEV, quiet, cfg_scale=2.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"C3-shaped DDIM step (128^2, cond 10, CFG on = 2 U-Net passes, batch {n}, fp32): first call {(t1-t0)/ (steps+1)*1e3:.1f} ms/step "
          f"(incl. capture), second {(t2-t1)/(steps+1)*1e3:.1f} ms/step", flush=True)
    net.set_inference_dtype(torch.bfloat16)
    S.ddim_sampling(net, sched, noise.clone(), 1, 1000, None, cond, 1000 // steps, DEV, quiet, cfg_scale=2.0)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    S.ddim_sampling(net, sched, noise.clone(), 1, 1000, None, cond, 1000 // steps, DEV, quiet, cfg_scale=2.0)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"C3-shaped DDIM step, bf16 storage: {(t4-t3)/(steps+1)*1e3:.1f} ms/step "
          f"({21.5e12/((t4-t3)/(steps+1))/1e12:.0f} algorithmic TFLOP/s)", flush=True)


def train(name, cfg, res, n, sr=None):
    torch.manual_seed(0)
    net = U_Net(**cfg).to(DEV)
    tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-5, seed=1, sr=sr)
    x0 = torch.randn(n, 3, res, res, device=DEV)
    t = timeit(lambda: tr.step(x0), warm=2, reps=4)
    print(f"{name}: {t*1e3:.1f} ms/step, {1/t:.2f} steps/s, {n/t:.1f} images/s", flush=True)
    del tr, net
    torch.cuda.empty_cache()


def main():
    what = sys.argv[1:] or ["attn", "c3", "c4", "c5"]
    if "attn" in what:
        attn()
    if "c4" in what:
        train("C4-shaped per-GPU training step (128^2, batch 32, fp32)", STANDARD, 128, 32)
    if "c5" in what:
        train("C5-shaped per-GPU SR training step (64->256, in_channel 6, tanh head, batch 8, fp32)",
              dict(STANDARD, in_channel=6, image_recon=True), 256, 8, sr=(64, 256, 100))
    if "c3" in what:
        c3()


if __name__ == "__main__":
    main()
