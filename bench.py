#!/usr/bin/env python3
"""Benchmark of the U-Net denoise step on MI355X (BASELINE.json metric: "U-Net denoise steps/sec").

    python bench.py --gpus N --steps K --warmup W

Headline (`value`): a *step* is one training pass of the hot path (q-sample -> U-Net forward -> MSE -> backward ->
Adam, plus the RCCL gradient all-reduce when N > 1) over one synthetic batch of BASELINE.json configs[1] ("same 64x64
U-Net on 1xMI355X, fp32"): standard architecture of SURVEY.md §8d (127.3 M parameters), per-GPU batch 4 (weak scaling:
global batch 4*N).  `value` = steps all ranks completed / max-over-ranks wall time, inputs resident in HBM.

The same JSON line also carries
  * `roofline`: the dominant contraction kernel class of that step, HIP events on the launch stream (by_class: MFMA and
    HBM fraction of every class);
  * `configs`: the other BASELINE configurations on this GPU, each with ms/step, steps/s and BOTH roofline fractions
    (SURVEY §8d "always print both"): c2 DDPM sampling at N=1 / N=4 (fp32 and bf16 storage), c3 DDIM + classifier-free
    guidance at 128x128 in bf16, c4 per-GPU training step at 128x128 batch 32, c5 SR 64->256 training step at batch 8;
  * `cpu_baseline`: the oracle (stock PyTorch-CPU ops = the reference's install_cpu_requirements.sh path) on the host
    cores, BASELINE.md §3: 1 warm-up + 10 timed training steps, N=1 and N=4 forward (sampling-step) rates;
  * under WORLD_SIZE > 1: `multi_gpu` (RCCL version, ranks, bus GB/s of one 64 MB bucket, all-reduce exposed ms, and
    the C4 shape -- BASELINE's scaling config -- as c4_steps_per_s).
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
# SURVEY.md §8d (exact walk of the module graph): forward FLOPs per image, activation elements per image, live forward
# parameter elements
FWD_FLOPS_PER_IMAGE = {64: 41_560_047_616, 128: 168_129_996_800, 256: 703_732_645_888}
FWD_FLOPS_PER_IMAGE_64 = FWD_FLOPS_PER_IMAGE[64]
ACT_ELEMS_PER_IMAGE = {64: 16_160_512, 128: 64_578_826, 256: 258_430_720}
LIVE_PARAMS = 122_192_131
PEAK_FP32_MFMA_TFLOPS = 157.3                    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0                   # dense bf16 MFMA (spec; AMD's 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBPS = 8000.0                           # HBM3E spec (6.3 TB/s is what a float4 copy reaches)


def alg_bytes(n, res, b, train, extra_params=0):
    """Algorithmic HBM bytes of one step: forward = b*(W + N*A); training = 3*act + 2*weights(b) + fp32
    weight-gradient write (4 B/param) + Adam 28 B/param."""
    w, a = LIVE_PARAMS + extra_params, ACT_ELEMS_PER_IMAGE[res] * n
    return b * (w + a) if not train else 3 * b * a + 2 * b * w + 4 * w + 28 * w


def fractions(ms, flops, nbytes, peak_tflops):
    t = ms * 1e-3
    return dict(ms_per_step=ms, steps_per_s=1e3 / ms, tflops_algorithmic=flops / t / 1e12,
                flop_frac=flops / t / 1e12 / peak_tflops, flop_peak_tflops=peak_tflops,
                hbm_GBps_algorithmic=nbytes / t / 1e9, byte_frac=nbytes / t / 1e9 / PEAK_HBM_GBPS)


def host_cores():
    """Threads for the CPU baseline: the cores this job may actually use = min(affinity mask, cgroup CPU quota)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return avail, quota, model


def cpu_baseline(batch, res):
    """oracle (plain torch CPU ops == the reference's CPU path) on the host cores: 1 warm-up + 10 timed
    training steps of C1, then 1 warm-up + up to 20 timed forwards at N=1 and N=4 (the sampling step), each bounded by a
    wall-clock budget so that the default bench run stays within minutes."""
    from oracle import unet_oracle as O
    avail, quota, model = host_cores()
    cap = int(os.environ.get("SDM_BENCH_CPU_THREADS", "0"))
    # a 1-GPU box hands this job a 16-CPU share of a 256-thread host (gpurun: "16 for one GPU"); when no cgroup quota says
    # otherwise, more threads than that only oversubscribe the share
    cores = cap if cap > 0 else max(1, min(avail, quota if quota else 16))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {model}; os.cpu_count={os.cpu_count()} affinity={avail} cgroup_quota={quota} "
          f"threads={cores}", file=sys.stderr, flush=True)
    torch.manual_seed(0)
    from models.U_Net import U_Net
    net = U_Net(**STANDARD)
    P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
    del net
    cfg = O.full_cfg(STANDARD)
    tr = O.OracleTrainer(P, cfg, O.CosineSchedule(1000), lr=2e-5)
    g = torch.Generator().manual_seed(1234)
    x0 = torch.randn(batch, 3, res, res, generator=g)
    eps = torch.randn(batch, 3, res, res, generator=g)
    t = torch.randint(1, 1000, (batch,), generator=g)

    def timed(fn, max_n, budget_s):
        tw = time.perf_counter()
        fn()                                                       # warm-up
        tw = time.perf_counter() - tw
        n = max(1, min(max_n, int(budget_s / max(tw, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return n, (time.perf_counter() - t0) / n

    n_tr, s_tr = timed(lambda: tr.step(x0, t, eps), 10, 14.0)
    print(f"[bench] cpu_baseline: train step {s_tr:.3f}s x{n_tr}", file=sys.stderr, flush=True)
    fw = {}
    with torch.no_grad():
        for n in (1, 4):
            xs, ts = x0[:n], t[:n]
            k, s = timed(lambda: O.unet_forward(P, cfg, xs, ts, None), 20, 5.0)
            fw[n] = (k, s)
    return dict(value=1.0 / s_tr, unit="U-Net train steps/s (batch %d, %dx%d, fwd+bwd+Adam)" % (batch, res, res),
                cores=cores, kind="port", cpu_model=model, torch=torch.__version__,
                threads_policy="min(affinity=%d, cgroup quota=%s, 16-CPU share of a 1-GPU box)" % (avail, quota),
                sample="%d timed C1 training steps after 1 warm-up of the oracle (stock torch CPU ops), same architecture / "
                       "batch; forward: %d (N=1) and %d (N=4) timed calls after 1 warm-up" % (n_tr, fw[1][0], fw[4][0]),
                sample_steps_per_s_n1=1.0 / fw[1][1], sample_steps_per_s_n4=1.0 / fw[4][1])


def time_gpu(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def other_configs(dev, sched, net64):
    """The other BASELINE configurations on this one GPU (inputs resident in HBM, synthetic Gaussian images, random-init
    weights).  Each entry: ms/step, steps/s, algorithmic TFLOP/s and GB/s, and both roofline fractions."""
    import diffusion_sampling_algorithms as S
    from models.U_Net import U_Net
    from sdm_hip import lib
    from sdm_hip.trainer import DiffusionTrainer
    quiet = lambda *a, **k: None
    out = {}
    g = torch.Generator(device=dev).manual_seed(4321)

    # ---- C2 sampling: DDPM hipGraph step at 64x64, N=1 and N=4, fp32 and bf16 storage (T=1000 sampler; K steps timed)
    def sample_ms(net, n, steps):
        xT = torch.randn(n, 3, 64, 64, device=dev, generator=g)
        S.ddpm_sampling(net, sched, xT, 1, 8, device="cuda", log=quiet)                  # warm kernels / scratch sizes
        torch.cuda.synchronize()
        # library launches of ONE denoise step: eager runs of two and of one step differ by exactly one step (the per-run
        # setup -- the modulation tables of the whole run -- is in both)
        counts = []
        for k in (1, 2):
            l0 = lib.launch_count()
            S.ddpm_sampling(net, sched, xT, 1, k, device="cuda", log=quiet, use_graph=False)
            counts.append(lib.launch_count() - l0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.ddpm_sampling(net, sched, xT, 1, steps, device="cuda", log=quiet)              # includes one graph capture
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        return ms, counts[1] - counts[0]

    for n in (1, 4):
        for dt, b, peak in ((torch.float32, 4, PEAK_FP32_MFMA_TFLOPS), (torch.bfloat16, 2, PEAK_BF16_MFMA_TFLOPS)):
            net64.set_inference_dtype(dt)
            ms, launches = sample_ms(net64, n, 1000)
            rec = fractions(ms, FWD_FLOPS_PER_IMAGE[64] * n, alg_bytes(n, 64, b, False), peak)
            rec.update(dtype="f32" if b == 4 else "bf16", batch=n, launches_per_step=launches,
                       what="DDPM step (U-Net forward + update) replayed from a hipGraph: the whole T=1000 sampling run incl. its "
                            "one capture (BASELINE configs[1] 'DDPM T=1000 sample'), per step")
            out["c2_sample_n%d_%s" % (n, "fp32" if b == 4 else "bf16")] = rec
    net64.set_inference_dtype(torch.float32)

    # ---- C3: 128x128, cond_dim=10, bf16, DDIM step with classifier-free guidance (2N batch), N=64
    torch.manual_seed(0)
    net = U_Net(**dict(STANDARD, cond_dim=10)).to(dev).eval()
    n = 64
    xT = torch.randn(n, 3, 128, 128, device=dev, generator=g)
    lab = torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(1237)), 10).float().to(dev)
    net.set_inference_dtype(torch.bfloat16)
    S.ddim_sampling(net, sched, xT, 1, 1000, None, lab, 500, dev, quiet, cfg_scale=3.0)     # warm (3 U-Net calls)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S.ddim_sampling(net, sched, xT, 1, 1000, None, lab, 20, dev, quiet, cfg_scale=3.0)      # 51 U-Net calls, incl. capture
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 51
    rec = fractions(ms, FWD_FLOPS_PER_IMAGE[128] * 2 * n, alg_bytes(2 * n, 128, 2, False, 793_600), PEAK_BF16_MFMA_TFLOPS)
    rec.update(dtype="bf16", batch=n, unet_batch=2 * n, images_per_s=n * 1e3 / ms,
               what="DDIM step with classifier-free guidance w=3 (one 2N-image U-Net pass), hipGraph, 51 steps incl. capture")
    out["c3_ddim_bf16_cfg"] = rec
    del net, xT
    torch.cuda.empty_cache()

    # ---- C5 sampling: the SR cold-diffusion sampler at 256x256 (generate_sr_images_diffusion.py:141-238,
    # diffusion_sampling_algorithms.py:150-217): 6-channel input (x_t | noised low-res condition), Tanh head, N=8, hipGraph
    torch.manual_seed(0)
    net = U_Net(**dict(STANDARD, in_channel=6, image_recon=True)).to(dev).eval()
    n = 8
    xT = torch.randn(n, 3, 256, 256, device=dev, generator=g)
    cond = torch.randn(n, 3, 256, 256, device=dev, generator=g)
    for dt, b, peak in ((torch.bfloat16, 2, PEAK_BF16_MFMA_TFLOPS), (torch.float32, 4, PEAK_FP32_MFMA_TFLOPS)):
        net.set_inference_dtype(dt)
        S.cold_diffusion_sampling(net, sched, xT, xT, 1, 1000, cond, None, 500, dev, quiet)          # warm (3 U-Net calls)
        torch.cuda.synchronize()
        nsteps = 51 if b == 2 else 11
        t0 = time.perf_counter()
        S.cold_diffusion_sampling(net, sched, xT, xT, 1, 1000, cond, None, 20 if b == 2 else 100, dev, quiet)   # incl. capture
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / nsteps
        rec = fractions(ms, FWD_FLOPS_PER_IMAGE[256] * n, alg_bytes(n, 256, b, False, 3_456), peak)
        rec.update(dtype="bf16" if b == 2 else "f32", batch=n, images_per_s=n * 1e3 / ms,
                   what="SR cold-diffusion sampling step at 256x256 (U-Net forward on cat(x_t, condition) + fused update), "
                        "hipGraph, %d steps incl. capture" % nsteps)
        out["c5_sr_sample_256" + ("_bf16" if b == 2 else "")] = rec
    del net, xT, cond
    torch.cuda.empty_cache()

    # ---- C4 / C5 per-GPU training steps (fp32; and the bf16 training path when the trainer offers it)
    def train_ms(cfg, res, n, sr, dtype):
        torch.manual_seed(0)
        net = U_Net(**cfg).to(dev)
        kw = {} if dtype == "f32" else {"train_dtype": torch.bfloat16}
        tr = DiffusionTrainer(net, sched, lr=2e-5, seed=1, sr=sr, **kw)
        x0 = torch.randn(n, 3, res, res, device=dev, generator=g)
        tr.step(x0)
        torch.cuda.synchronize()
        l0 = lib.launch_count()
        tr.step(x0)
        launches = lib.launch_count() - l0
        ms = time_gpu(lambda: tr.step(x0), 1, 4)
        del tr, net, x0
        torch.cuda.empty_cache()
        return ms, launches

    import inspect
    has_bf16_train = "train_dtype" in inspect.signature(DiffusionTrainer.__init__).parameters
    for key, cfg, res, n, sr, extra in (
            ("c4_train_pergpu", STANDARD, 128, 32, None, 0),
            ("c5_sr_train_256", dict(STANDARD, in_channel=6, image_recon=True), 256, 8, (64, 256, 100), 3_456)):
        for dtype in (("f32", "bf16") if has_bf16_train else ("f32",)):
            ms, launches = train_ms(cfg, res, n, sr, dtype)
            b, peak = (4, PEAK_FP32_MFMA_TFLOPS) if dtype == "f32" else (2, PEAK_BF16_MFMA_TFLOPS)
            rec = fractions(ms, 3 * FWD_FLOPS_PER_IMAGE[res] * n, alg_bytes(n, res, b, True, extra), peak)
            rec.update(dtype=dtype, batch=n, images_per_s=n * 1e3 / ms, launches_per_step=launches,
                       what="per-GPU training step (q-sample, forward, MSE, backward, Adam), 4 steps timed")
            out[key + ("" if dtype == "f32" else "_bf16")] = rec
    # ---- the N=1 point of C4's strong-scaling curve (global batch 256 on one GPU); N>1 points: multi_gpu.c4_strong
    out["c4_strong_global256"] = c4_strong_scaling(dev, sched, 1)
    return out


def multi_gpu_report(dev, world, rank, sched, tr64, x64):
    """Multi-GPU readiness fields: what RCCL sees, one bucket's bus bandwidth, how much of the all-reduce is exposed, and
    BASELINE's scaling configuration (C4: 128x128, 32 images per GPU)."""
    from models.U_Net import U_Net
    from sdm_hip.trainer import DiffusionTrainer
    rep = {"backend": dist.get_backend(), "nranks": dist.get_world_size()}
    try:
        rep["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rep["rccl_version"] = None
    buf = torch.zeros(16 << 20, dtype=torch.float32, device=dev)        # one 64 MB bucket
    for _ in range(2):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    rep["allreduce_64MB_ms"] = 1e3 * dt
    rep["allreduce_64MB_busGBps"] = 2.0 * (world - 1) / world * buf.numel() * 4 / dt / 1e9
    del buf
    # exposed all-reduce: the same step with the collective switched off (gradients stay local; timing only)
    def steps_ms(k=8):
        dist.barrier()
        return time_gpu(lambda: tr64.step(x64), 1, k)
    with_ar = steps_ms()
    tr64.reducer.world = 1
    without = steps_ms()
    tr64.reducer.world = world
    rep["step_ms_with_allreduce"], rep["step_ms_without_allreduce"] = with_ar, without
    rep["allreduce_exposed_ms"] = with_ar - without
    # C4 shape under data parallelism (BASELINE configs[3]: 128x128, 32 images per GPU), fp32 and the bf16 training path
    for tag, kw in (("", {}), ("_bf16", {"train_dtype": torch.bfloat16})):
        torch.manual_seed(0)
        net = U_Net(**STANDARD).to(dev)
        tr = DiffusionTrainer(net, sched, lr=2e-5, seed=77, **kw)
        x0 = torch.randn(32, 3, 128, 128, device=dev)
        dist.barrier()
        ms = time_gpu(lambda: tr.step(x0), 2, 4)
        tm = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        rep["c4%s_ms_per_step" % tag] = float(tm)
        rep["c4%s_steps_per_s" % tag] = 1e3 / float(tm)
        rep["c4%s_images_per_s" % tag] = world * 32 * 1e3 / float(tm)
        del tr, net, x0
        torch.cuda.empty_cache()
    rep["c4_config"] = "128x128, 32 images per GPU, DDPM training, global batch %d; fp32 and bf16 storage" % (32 * world)
    rep["c4_strong"] = c4_strong_scaling(dev, sched, world)
    rep["dp_backend"], rep["dp_wire"] = tr64.exchange.backend, tr64.exchange.wire
    return rep


def c4_strong_scaling(dev, sched, world):
    """BASELINE configs[3] as SURVEY §8d defines it: GLOBAL batch 256 at 128x128, fixed; every rank trains 256 / world images
    per step ("256x1, 128x2, 64x4, 32x8").  The N=1 line of that curve is the same function called with world == 1 from
    other_configs().  Time = max over ranks; images_per_s = 256 / that."""
    from models.U_Net import U_Net
    from sdm_hip.trainer import DiffusionTrainer
    out = {"global_batch": 256, "per_gpu_batch": 256 // world, "scaling": "strong"}
    for tag, kw in (("bf16", {"train_dtype": torch.bfloat16}), ("f32", {})):
        torch.manual_seed(0)
        net = tr = x0 = None
        try:
            net = U_Net(**STANDARD).to(dev)
            tr = DiffusionTrainer(net, sched, lr=2e-5, seed=78, **kw)
            x0 = torch.randn(256 // world, 3, 128, 128, device=dev)
            torch.cuda.reset_peak_memory_stats(dev)
            if world > 1:
                dist.barrier()
            ms = time_gpu(lambda: tr.step(x0), 1, 3)
            if world > 1:
                tm = torch.tensor([ms], device=dev, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                ms = float(tm)
            out[tag] = {"ms_per_step": ms, "images_per_s": 256 * 1e3 / ms, "peak_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
        except torch.OutOfMemoryError:
            out[tag] = {"ms_per_step": None, "note": "%d images per GPU do not fit" % (256 // world)}
        del tr, net, x0
        torch.cuda.empty_cache()
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: THIS process becomes the launcher.  It starts N fresh
    interpreters of this file, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly
    what `python -m torch.distributed.run --nproc-per-node N` would set), relays rank 0's JSON line (rank 0 inherits
    stdout; the other ranks' stdout goes to stderr) and exits with the ranks' return code.  The launcher itself never
    touches the GPU (no torch.cuda call, no library import) and never replaces itself with another program."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # one rank failed: the others would wait in a collective forever
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch")
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SDM_BENCH_DRYRUN") == "1":
        # launcher rehearsal without a GPU (tests/test_abi_and_host.py): the ranks rendezvous over gloo, count themselves
        # with an all-reduce and rank 0 prints what the communicator saw
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        one = torch.ones(1)
        dist.all_reduce(one)
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": dist.get_world_size(), "ranks_counted": int(one.item()),
                              "gpus_arg": args.gpus}))
        dist.destroy_process_group()
        return
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
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from models.U_Net import U_Net
    import degraders
    from sdm_hip import lib
    from sdm_hip import ops as _ops
    from sdm_hip.trainer import DiffusionTrainer

    torch.manual_seed(0)
    net = U_Net(**STANDARD).to(dev)
    sched = degraders.CosineNoiseDegradation(1000)
    tr = DiffusionTrainer(net, sched, lr=2e-5, seed=1235)               # the trainer offsets the seed by the rank
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x0 = torch.randn(args.batch, 3, args.res, args.res, device=dev, generator=g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(x0)
    barrier()
    l0 = lib.launch_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(x0)
    t_enq = time.perf_counter() - t0           # host time to enqueue K steps (GPU still running)
    barrier()
    dt = time.perf_counter() - t0
    launches_per_step = (lib.launch_count() - l0) / args.steps
    # what the shipped CLI loop pays: one float(loss) read-back per step (train_diffusion.py:354,366 read it twice), so the
    # host cannot run ahead of the GPU
    barrier()
    tl0 = time.perf_counter()
    for _ in range(max(1, args.steps // 2)):
        float(tr.step(x0))
    barrier()
    step_with_loss_ms = 1e3 * (time.perf_counter() - tl0) / max(1, args.steps // 2)
    # roofline leg: the same step, eager, with a HIP event pair around every contraction launch on its launch stream
    # (a replayed hipGraph cannot carry per-kernel events, so this is a second, shorter region of the same workload);
    # weight-gradient kernels normally overlap data-gradient kernels on a second stream: for the per-kernel numbers that
    # overlap is switched off so each launch has the chip to itself (same kernels, same shapes)
    graphed = tr.use_graph
    tr.use_graph = False
    side_was = _ops.SIDE.enabled
    _ops.SIDE.enabled = False
    prof_steps = max(1, min(args.steps, 10))
    tr.step(x0)
    barrier()
    # what an event pair adds to the launch it brackets (median over an empty kernel; still contains that kernel's own ~1-2 us)
    bracket_ms = lib.prof_bracket_ms(_ops._stream(), 64)
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

    multi = None
    if world > 1:
        multi = multi_gpu_report(dev, world, rank, sched, tr, x0)
        dist.barrier()
    if rank != 0:
        dist.destroy_process_group()
        return

    configs = None
    if world == 1 and not args.no_configs:
        del tr
        configs = other_configs(dev, sched, net.eval())
        for k, v in configs.items():
            if "ms_per_step" in v:
                print(f"[bench] {k}: {v['ms_per_step']:.3f} ms/step, flop_frac {v['flop_frac']:.3f}, byte_frac {v['byte_frac']:.3f}",
                      file=sys.stderr, flush=True)
            else:
                print(f"[bench] {k}: {json.dumps(v)}", file=sys.stderr, flush=True)

    # HBM bytes per launch of each class from the committed rocprofv3 PMC passes of this same command
    # (profiles/*_hbm_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); None when the file is absent
    traffic, traffic_file = {}, None
    for name in ("r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = {k: v["total"] for k, v in json.load(f)["per_class_hbm_bytes_per_launch"].items()}
            traffic_file = name
            break
        except Exception:
            pass
    # Per-launch durations: the time between a launch's two HIP events brackets a few us more than the kernel (dispatch of the
    # kernel after the first event, completion signalling before the second).  That overhead is measured live (the same pair
    # around an EMPTY kernel, median of 64) and removed, leaving the empty kernel 2 us of its own so that the correction stays
    # on the conservative side of rocprofv3's kernel durations (profiles/r03_train_step_profiled_leg.txt: 34.2 us per launch
    # of the dominant class where this gives 35.0 and the raw events 38.9).  The raw figures are reported beside it.
    over_ms = max(0.0, bracket_ms - 0.002)
    for v in prof.values():
        v["ms_raw"] = v["ms"]
        v["ms"] = max(0.0, v["ms"] - over_ms * v["launches"])
    dom = max(prof, key=lambda k: prof[k]["ms"])
    d = prof[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    achieved_raw = d["flops"] / (d["ms_raw"] * 1e-3) / 1e12 if d["ms_raw"] > 0 else 0.0
    step_flops = 3 * FWD_FLOPS_PER_IMAGE_64 * args.batch * (args.res / 64) ** 2
    step_ms = 1e3 * dt / args.steps
    out = {
        "metric": "U-Net denoise steps/sec (train fwd+bwd+Adam)",
        "value": world * args.steps / dt,
        "unit": "train steps/s (one step = one U-Net fwd+bwd+Adam pass over a batch of %d %dx%d images on one GPU)"
                % (args.batch, args.res, args.res),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: standard 127.3M-param U-Net (L=4,R=1,attn=[2,3],D=512,ch 128..512), "
                               "%dx%d, unconditional, cosine schedule, fp32, DDPM noise-prediction training" % (args.res, args.res),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "parallelism": "dp%d" % world if world > 1 else "single"},
        "loss": loss_v,
        "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps,
        "launches_per_step": launches_per_step,
        "step_with_loss_ms": step_with_loss_ms,
        "train_step_hipgraph": bool(graphed),
        "step_tflops_algorithmic": step_flops / (dt / args.steps) / 1e12,
        "step_flop_frac": step_flops / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "step_byte_frac": alg_bytes(args.batch, args.res, 4, True) / (dt / args.steps) / 1e9 / PEAK_HBM_GBPS,
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic.get(dom),
                     "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; bytes per "
                                        "launch)" % traffic_file) if dom in traffic else None,
                     "algorithmic_bytes_per_launch": d["bytes"] / max(1, d["launches"]),
                     "hbm_GBps_algorithmic": d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0,
                     "launches": d["launches"], "avg_launch_ms": d["ms"] / max(1, d["launches"]),
                     "event_bracket_ms": bracket_ms, "avg_launch_ms_raw_events": d["ms_raw"] / max(1, d["launches"]),
                     "achieved_raw_events": achieved_raw, "frac_raw_events": achieved_raw / PEAK_FP32_MFMA_TFLOPS,
                     "measured_over": "%d steps (%.2f ms/step) with HIP event pairs around every contraction launch, minus the pair's "
                                      "own overhead (event_bracket_ms - 2 us, measured around an empty kernel; *_raw_events = without "
                                      "that correction), run right after the timed region with the weight-gradient side stream off "
                                      "so that each launch has the chip to itself (timed region: side stream %s, hipGraph %s)"
                                      % (prof_steps, 1e3 * dt_prof / prof_steps, "on" if side_was else "off", bool(graphed)),
                     "by_class": {k: {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["launches"] / prof_steps,
                                      "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0,
                                      "mfma_frac": (v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if v["ms"] > 0 else 0.0,
                                      "hbm_frac": (v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS) if v["ms"] > 0 else 0.0}
                                  for k, v in prof.items()}},
    }
    if configs is not None:
        out["configs"] = configs
        # kept for continuity with round 1's line (batch-4 DDPM hipGraph step)
        out["sample_steps_per_s"] = configs["c2_sample_n4_fp32"]["steps_per_s"]
        out["sample_steps_per_s_bf16"] = configs["c2_sample_n4_bf16"]["steps_per_s"]
    if multi is not None:
        out["multi_gpu"] = multi
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.batch, args.res)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
