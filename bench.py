#!/usr/bin/env python3
"""Benchmark of the U-Net denoise step on MI355X (BASELINE.json metric: "U-Net denoise steps/sec").

    python bench.py --gpus N --steps K --warmup W

A *step* is one training pass of the hot path (q-sample -> U-Net forward -> MSE -> backward -> Adam,
plus the RCCL gradient all-reduce when N > 1) over one synthetic batch: BASELINE.json configs[1]
("same 64x64 U-Net on 1xMI355X, fp32"), standard architecture of SURVEY.md §8d (127.3 M parameters),
per-GPU batch 4 (weak scaling: the global batch is 4*N).  `value` = steps all ranks completed / max-over-
ranks wall time, inputs resident in HBM.  The same JSON line also carries the DDPM sampling rate of the
hipGraph-captured step (`sample_steps_per_s`), the roofline of the dominant kernel class measured with HIP
events on the launch stream inside the timed region, and a CPU baseline (the oracle = stock PyTorch-CPU
ops, the reference's install_cpu_requirements.sh path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "simple-diffusion-model_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

STANDARD = dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=512, cond_dim=None, num_layers=4,
                attn_layers=[2, 3], num_heads=1, dim_per_head=None, min_channel=128, max_channel=512, image_recon=False)
FWD_FLOPS_PER_IMAGE_64 = 41_560_047_616          # SURVEY.md §8d (exact walk)
PEAK_FP32_MFMA_TFLOPS = 157.3                    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (spec)


def cpu_baseline(batch, res, steps=3):
    """Oracle (plain torch CPU ops == the reference's CPU path) training steps on the host cores."""
    from oracle import unet_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))       # a 1-GPU box gives this job a 16-CPU share; more threads only oversubscribe
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: os.cpu_count={os.cpu_count()} affinity={avail} threads={cores}", file=sys.stderr, flush=True)
    torch.manual_seed(0)
    from models.U_Net import U_Net
    net = U_Net(**STANDARD)
    P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
    del net
    tr = O.OracleTrainer(P, O.full_cfg(STANDARD), O.CosineSchedule(1000), lr=2e-5)
    g = torch.Generator().manual_seed(1234)
    x0 = torch.randn(batch, 3, res, res, generator=g)
    eps = torch.randn(batch, 3, res, res, generator=g)
    t = torch.randint(1, 1000, (batch,), generator=g)
    tw = time.perf_counter()
    tr.step(x0, t, eps)                                        # warm-up
    tw = time.perf_counter() - tw
    print(f"[bench] cpu_baseline: warm-up step {tw:.2f}s", file=sys.stderr, flush=True)
    steps = max(1, min(steps, int(20.0 / max(tw, 1e-3))))      # bound the sample to ~20 s of CPU work
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x0, t, eps)
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit="U-Net train steps/s (batch %d, %dx%d, fwd+bwd+Adam)" % (batch, res, res),
                cores=cores, kind="port",
                sample="%d timed steps after 1 warm-up of the oracle (stock torch %s CPU ops), same architecture/batch"
                       % (steps, torch.__version__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch")
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--sample-steps", type=int, default=50, help="DDPM steps timed for the sampling rate (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(1, torch.cuda.device_count())     # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SDM_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from models.U_Net import U_Net
    import degraders
    import diffusion_sampling_algorithms as S
    from sdm_hip import lib
    from sdm_hip.trainer import DiffusionTrainer

    torch.manual_seed(0)
    net = U_Net(**STANDARD).to(dev)
    sched = degraders.CosineNoiseDegradation(1000)
    tr = DiffusionTrainer(net, sched, lr=2e-5, seed=1235 + rank)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x0 = torch.randn(args.batch, 3, args.res, args.res, device=dev, generator=g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(x0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(x0)
    t_enq = time.perf_counter() - t0           # host time to enqueue K steps (GPU still running)
    barrier()
    dt = time.perf_counter() - t0
    # roofline leg: the same step, eager, with a HIP event pair around every contraction launch on its launch stream
    # (a replayed hipGraph cannot carry per-kernel events, so this is a second, shorter region of the same workload)
    # (b) weight-gradient kernels normally overlap data-gradient kernels on a second stream; for the per-kernel
    # numbers that overlap is switched off so each launch has the chip to itself (same kernels, same shapes)
    from sdm_hip import ops as _ops
    graphed = tr.use_graph
    tr.use_graph = False
    side_was = _ops.SIDE.enabled
    _ops.SIDE.enabled = False
    prof_steps = max(1, min(args.steps, 10))
    tr.step(x0)
    barrier()
    lib.prof_begin(200000)
    tp0 = time.perf_counter()
    for _ in range(prof_steps):
        tr.step(x0)
    barrier()
    dt_prof = time.perf_counter() - tp0
    prof = lib.prof_end()
    tr.use_graph = graphed
    _ops.SIDE.enabled = side_was
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss_v = float(loss)
    print(f"[bench] rank {rank}: {args.steps} train steps in {dt:.3f}s", file=sys.stderr, flush=True)

    # ---- sampling rate of the hipGraph-captured DDPM step (rank 0 only, replicas need no collective)
    sample_rate = sample_rate_bf16 = None
    if args.sample_steps > 0 and rank == 0:
        xT = torch.randn(args.batch, 3, args.res, args.res, device=dev, generator=g)
        quiet = lambda *a, **k: None
        S.ddpm_sampling(net, sched, xT, 1, 4, device="cuda", log=quiet)                 # warm the graph path
        torch.cuda.synchronize()
        ts = time.perf_counter()
        S.ddpm_sampling(net, sched, xT, 1, args.sample_steps, device="cuda", log=quiet)
        torch.cuda.synchronize()
        sample_rate = args.sample_steps / (time.perf_counter() - ts)     # includes one capture; amortised over T
        # same sampler with the U-Net in bf16 storage (fp32 accumulation / statistics / sampler state)
        net.set_inference_dtype(torch.bfloat16)
        S.ddpm_sampling(net, sched, xT, 1, 4, device="cuda", log=quiet)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        S.ddpm_sampling(net, sched, xT, 1, args.sample_steps, device="cuda", log=quiet)
        torch.cuda.synchronize()
        sample_rate_bf16 = args.sample_steps / (time.perf_counter() - ts)
        net.set_inference_dtype(torch.float32)
        print(f"[bench] sampling: {sample_rate:.1f} steps/s fp32, {sample_rate_bf16:.1f} steps/s bf16", file=sys.stderr, flush=True)

    if world > 1:
        dist.barrier()                      # rank 0 may still have been sampling
    if rank != 0:
        dist.destroy_process_group()
        return

    # HBM bytes per launch of each class from the committed rocprofv3 PMC passes of this same command
    # (profiles/r01_hbm_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); None when the file is absent
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            traffic = {k: v["total"] for k, v in json.load(f)["per_class_hbm_bytes_per_launch"].items()}
    except Exception:
        pass
    dom = max(prof, key=lambda k: prof[k]["ms"])
    d = prof[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    step_flops = 3 * FWD_FLOPS_PER_IMAGE_64 * args.batch * (args.res / 64) ** 2
    out = {
        "metric": "U-Net denoise steps/sec (train fwd+bwd+Adam)",
        "value": world * args.steps / dt,
        "unit": "train steps/s (one step = one U-Net fwd+bwd+Adam pass over a batch of %d %dx%d images on one GPU)"
                % (args.batch, args.res, args.res),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: standard 127.3M-param U-Net (L=4,R=1,attn=[2,3],D=512,ch 128..512), "
                               "%dx%d, unconditional, cosine schedule, fp32, DDPM noise-prediction training" % (args.res, args.res),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "parallelism": "dp%d" % world if world > 1 else "single"},
        "loss": loss_v,
        "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps,
        "train_step_hipgraph": bool(graphed),
        "step_tflops_algorithmic": step_flops / (dt / args.steps) / 1e12,
        "sample_steps_per_s": sample_rate,
        "sample_steps_per_s_bf16": sample_rate_bf16,
        "sample_config": None if sample_rate is None else "DDPM hipGraph step, batch %d, %d steps timed incl. capture"
                                                          % (args.batch, args.sample_steps),
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic.get(dom),
                     "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; "
                                       "bytes per launch)" if dom in traffic else None,
                     "algorithmic_bytes_per_launch": d["bytes"] / max(1, d["launches"]),
                     "hbm_GBps_algorithmic": d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0,
                     "launches": d["launches"], "avg_launch_ms": d["ms"] / max(1, d["launches"]),
                     "measured_over": "%d steps (%.2f ms/step) with HIP event pairs around every contraction launch, run right "
                                      "after the timed region with the weight-gradient side stream off so that each launch "
                                      "has the chip to itself (timed region: side stream %s, hipGraph %s)"
                                      % (prof_steps, 1e3 * dt_prof / prof_steps, "on" if side_was else "off", bool(graphed)),
                     "by_class": {k: {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["launches"] / prof_steps,
                                      "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0}
                                  for k, v in prof.items()}},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.batch, args.res)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
