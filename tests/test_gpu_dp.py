"""Data-parallel training path end to end on ONE GPU: two ranks (gloo backend on CUDA tensors, both on cuda:0)
each train on half of a batch through DiffusionTrainer (bucketed all-reduce hooked into the backward, per-bucket
fused Adam with grad_scale = 1/world); the result must equal a single-rank run on the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NARROW = dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3, num_layers=2,
              attn_layers=[1], num_heads=2, dim_per_head=16, min_channel=32, max_channel=64)


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "simple-diffusion-model_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _data():
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 4, 3, 16, 16, generator=g)
    eps = torch.randn(2, 4, 3, 16, 16, generator=g)
    t = torch.randint(1, 1000, (2, 4), generator=g)
    lab = torch.eye(3)[torch.randint(0, 3, (2, 4), generator=g)]
    return x0, eps, t, lab


def _train(world, rank, pg_ready, bf16=False):
    _setup_paths()
    import degraders as D
    from models.U_Net import U_Net
    from oracle import unet_oracle as O
    from sdm_hip.trainer import DiffusionTrainer
    cfg = O.full_cfg(NARROW)
    net = U_Net(**cfg)
    net.load_state_dict(O.seeded_params(cfg, seed=3, gain=1.8), strict=True)
    net = net.cuda()
    tr = DiffusionTrainer(net, D.CosineNoiseDegradation(1000), lr=2e-3, bucket_mb=1,
                          train_dtype=torch.bfloat16 if bf16 else None)
    assert tr.world == world
    x0, eps, t, lab = _data()
    per = 4 // world
    sl = slice(rank * per, (rank + 1) * per)
    losses = []
    for i in range(2):
        losses.append(float(tr.step(x0[i, sl].cuda(), lab[i, sl].cuda(), eps=eps[i, sl].cuda(), t=t[i, sl].cuda())))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().contiguous().clone() for k, v in net.state_dict().items()}, losses, len(tr.reducer.buckets)


def _worker(rank, world, port, out_path, bf16=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd, losses, nb = _train(world, rank, True, bf16)
        if rank == 0:
            torch.save({"sd": sd, "losses": losses, "buckets": nb}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True])
def test_two_rank_training_equals_single_rank(tmp_path, bf16):
    """bf16=True: the bf16 training path under data parallelism -- weight gradients and the all-reduced flat buffer stay
    fp32 and every image's activations are rounded independently of its batch neighbours, so the FIRST step of two ranks on
    half batches equals one rank's on the whole batch up to fp32 summation order; later steps agree statistically."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out, bf16), nprocs=2, join=True)
    dp = torch.load(out, weights_only=True)
    assert dp["buckets"] >= 2
    ref_sd, ref_losses, _ = _train(1, 0, False, bf16)
    if bf16:
        # After the first update the two runs' fp32 weights differ in their last bits (summation order); on bf16 rows such
        # a bit can flip an activation's rounding (a 2^-9 jump), and Adam's early steps (~lr*sign(g)) turn gradient entries
        # that sit at that noise level into +-lr moves.  So the bound here is statistical: almost every weight within a
        # fraction of one update, none further than a few updates.
        off = total = 0
        worst = 0.0
        for k, v in ref_sd.items():
            d = (dp["sd"][k] - v).abs()
            off += int((d > 2e-4).sum())
            total += d.numel()
            worst = max(worst, float(d.max()))
        assert off / total < 0.02 and worst <= 4 * 2e-3, (off, total, worst)      # (|m/sqrt(v)| can exceed 1 in early steps)
        return
    bad = []
    for k, v in ref_sd.items():
        # Adam's first updates are ~lr*sign(g): a gradient that differs in its last bits near zero moves a weight by up to
        # ~lr, so the bound is a fraction of lr (2e-3), not of the weight
        if not torch.allclose(dp["sd"][k], v, rtol=1e-3, atol=2e-4):
            bad.append((k, float((dp["sd"][k] - v).abs().max())))
    assert not bad, bad[:8]


# ---- the same property on the bench workload: 127 M parameters, five buckets of >= 64 MB, y_scale projections in the last one
def _train_std(world, rank):
    _setup_paths()
    import degraders as D
    from models.U_Net import U_Net
    from oracle import unet_oracle as O
    from sdm_hip.trainer import DiffusionTrainer
    torch.manual_seed(0)
    net = U_Net(**dict(O.STANDARD_CFG)).cuda()
    tr = DiffusionTrainer(net, D.CosineNoiseDegradation(1000), lr=1e-3)
    g = torch.Generator().manual_seed(17)
    x0 = torch.randn(2, 3, 64, 64, generator=g)
    eps = torch.randn(2, 3, 64, 64, generator=g)
    t = torch.tensor([640, 90])
    per = 2 // world
    sl = slice(rank * per, (rank + 1) * per)
    loss = float(tr.step(x0[sl].cuda(), eps=eps[sl].cuda(), t=t[sl].cuda()))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().contiguous().clone() for k, v in net.state_dict().items()}, loss, len(tr.reducer.buckets)


def _worker_std(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd, loss, nb = _train_std(world, rank)
        losses = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(losses, torch.tensor([loss]))
        if rank == 0:
            torch.save({"sd": sd, "losses": [float(v) for v in losses], "buckets": nb}, out_path)
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_rank_standard_architecture(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp_std.pt")
    mp.spawn(_worker_std, args=(2, port, out), nprocs=2, join=True)
    dp = torch.load(out, weights_only=True)
    assert dp["buckets"] >= 4
    ref_sd, ref_loss, _ = _train_std(1, 0)
    assert abs(sum(dp["losses"]) / 2 - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    off = total = 0
    for k, v in ref_sd.items():
        d = (dp["sd"][k] - v).abs()
        off += int((d > 1e-4).sum())       # lr = 1e-3: a tenth of one update
        total += d.numel()
    assert off / total < 1e-3, (off, total)


def test_bench_two_rank_rehearsal_reports_multi_gpu_fields(tmp_path):
    """`python bench.py --gpus 2` (gloo rehearsal, both ranks on the one GPU; the driver runs the same command with RCCL
    on 2/4/8 GPUs): the JSON line carries the multi-GPU readiness fields -- what the communicator sees (backend, ranks,
    RCCL version), the bus bandwidth of one 64 MB bucket, the all-reduce exposed per step and BASELINE's scaling
    configuration (C4: 128x128, 32 images per GPU) as c4_steps_per_s."""
    import json
    import subprocess
    env = dict(os.environ, SDM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    # exactly the driver's command: no external launcher, bench.py starts the two ranks itself
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 8
    m = out["multi_gpu"]
    assert m["nranks"] == 2 and m["backend"] == "gloo"
    for k in ("allreduce_64MB_ms", "allreduce_64MB_busGBps", "allreduce_exposed_ms", "step_ms_with_allreduce",
              "step_ms_without_allreduce", "c4_ms_per_step", "c4_steps_per_s", "c4_images_per_s", "c4_bf16_ms_per_step",
              "c4_bf16_steps_per_s"):
        assert isinstance(m[k], float) and m[k] == m[k], k
    assert m["c4_steps_per_s"] > 0 and "cpu_baseline" not in out


def test_library_owned_communicator_one_rank():
    """sdm_dp_* on the one GPU (RCCL communicator of size 1, owned by libsdm_hip.so): fp32 wire leaves the bucket as it is,
    the bf16 wire rounds it (pack -> fp32 "sum" of the owner's unrounded values -> one rounding -> unpack), odd lengths and
    the staging growth included; a dead handle is refused; DiffusionTrainer(dp_backend="sdm") steps exactly like the default."""
    _setup_paths()
    import degraders as D
    from models.U_Net import U_Net
    from oracle import unet_oracle as O
    from sdm_hip import dp, lib
    from sdm_hip.trainer import DiffusionTrainer
    assert lib.call("sdm_dp_available") > 20000
    ex = dp.SdmExchange(None, "f32")
    assert (ex.world, ex.rank) == (1, 0)
    assert lib.call("sdm_dp_world", ex.handle) == 1 and lib.call("sdm_dp_rank", ex.handle) == 0
    g = torch.Generator().manual_seed(9)
    for n in (8, 1003, 160000, 5_000_001):
        x = (torch.randn(n + 4, generator=g) * 3).cuda()[4 if n != 1003 else 3:]     # 16-byte aligned (not 64); one odd start
        y = x.clone()
        ex.all_reduce(y).wait()
        torch.cuda.synchronize()
        assert torch.equal(y, x), n
        ex.wire = "bf16"
        ex.all_reduce(y).wait()
        torch.cuda.synchronize()
        assert torch.equal(y, x.to(torch.bfloat16).float()), n
        ex.wire = "f32"
    ex.check()
    with pytest.raises(lib.SdmError, match="null buffer"):
        lib.call("sdm_dp_allreduce_bucket", ex.handle, None, 8, 0, 0, None)
    h = ex.handle
    ex.close()
    assert lib.raw("sdm_dp_world")(h) == -1 and "communicator" in lib.last_error()

    cfg = O.full_cfg(NARROW)
    x0, eps, t, lab = _data()
    states = []
    for kw in ({}, {"dp_backend": "sdm"}, {"dp_backend": "sdm", "dp_wire": "bf16"}):
        net = U_Net(**cfg)
        net.load_state_dict(O.seeded_params(cfg, seed=3, gain=1.8), strict=True)
        net = net.cuda()
        tr = DiffusionTrainer(net, D.CosineNoiseDegradation(1000), lr=2e-3, bucket_mb=1, **kw)
        loss = float(tr.step(x0[0].cuda(), lab[0].cuda(), eps=eps[0].cuda(), t=t[0].cuda()))
        torch.cuda.synchronize()
        tr.exchange.check()
        states.append((loss, {k: v.detach().clone() for k, v in net.state_dict().items()}))
        tr.exchange.close()
    assert all(abs(s[0] - states[0][0]) <= 1e-5 * abs(states[0][0]) for s in states)
    for k, v in states[0][1].items():                                  # fp32 wire, one rank: the same step up to the
        if v.dtype.is_floating_point:                                  # summation order of the backward's atomics
            assert float((v - states[1][1][k]).abs().mean()) < 0.05 * 2e-3, k
    # bf16 wire: gradients rounded to bf16 before Adam; the first Adam step is lr * g / (|g| + eps): rounding g moves it by
    # at most ~2^-9 relative of lr
    worst = max(float((v - states[2][1][k]).abs().max()) for k, v in states[0][1].items() if v.dtype.is_floating_point)
    assert worst <= 2e-3 * 2.0 ** -5, worst
