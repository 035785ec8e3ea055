"""Parity of every BASELINE.json configuration in the MODE it is named in:

  configs[0]/C1  64x64 batch 4, 10 Adam steps: the known-answer run of SURVEY.md §8c (losses and the final state_dict
                 checksum recorded from the reference's CPU path), no oracle needed at run time
  configs[2]/C3  128x128, cond_dim=10, bf16 storage, DDIM with classifier-free guidance, hipGraph + eager
  configs[3]/C4  128x128 DDPM training step (loss + all 148 live gradient tensors; attention backward at S=1024)
  configs[4]/C5  SR 64->256 training step, in_channel=6, Tanh head (fp32, and the bf16 training path)

plus the embedding broadcast cases of ADVICE r1 (t of shape [1] against labels [n, cond_dim] and the reverse).
All through the C ABI; the oracle (oracle/unet_oracle.py, pinned by tests/golden) is the checker.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import unet_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return unet_oracle


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def standard(O, **over):
    """Standard architecture under torch.manual_seed(0): the module on the GPU and its CPU parameter dict."""
    from models.U_Net import U_Net
    cfg = dict(O.STANDARD_CFG, **over)
    torch.manual_seed(0)
    net = U_Net(**cfg)
    P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
    return cfg, net.cuda(), P


def check_grads(O, cfg, net, grads, ref_g, lim=3e-4, bias_lim=2e-3):
    dead = set(O.dead_param_names(cfg))
    bad, n = [], 0
    for k, p in net.named_parameters():
        if k in dead:
            assert ref_g[k] is None
            continue
        n += 1
        r = rel(grads[p], ref_g[k])
        # conv biases in front of a GroupNorm: the true gradient is a cancelling sum ~1e-7 of its terms
        if r > (bias_lim if (".conv_block_" in k and k.endswith("conv_layer.0.bias")) else lim):
            bad.append((k, r))
    assert not bad, bad[:10]
    return n


# ------------------------------------------------------------------------------------------------ C1
C1_LOSSES = [1.00576866, 0.99192476, 0.99215859, 1.00265920, 0.99446535, 1.00586379, 0.99707270, 1.00778830, 1.01259565,
             1.00419104]
C1_INIT_SUM = 13053.625085999178
C1_FINAL_SUM = 13058.721764604354


def test_c1_ten_step_known_answer(dev):
    """SURVEY.md §8c protocol, verbatim: torch.manual_seed(0) init (checksum 13053.625...), Adam(2e-5, (0.5, 0.999)),
    cosine schedule, data = randn(40,3,64,64, G(1234)), eps from G(1235), t from G(1236) (step 0 -> [753,878,655,810]);
    per-step losses and the state_dict checksum after 10 steps, recorded from the reference's PyTorch-CPU path."""
    import degraders as D
    from models.U_Net import U_Net
    from oracle.unet_oracle import STANDARD_CFG
    from sdm_hip.trainer import DiffusionTrainer
    torch.manual_seed(0)
    net = U_Net(**STANDARD_CFG)
    s0 = sum(float(v.double().sum()) for v in net.state_dict().values())
    assert abs(s0 - C1_INIT_SUM) < 1e-6, s0
    net = net.cuda()
    tr = DiffusionTrainer(net, D.CosineNoiseDegradation(1000), lr=2e-5, betas=(0.5, 0.999))
    data = torch.randn(40, 3, 64, 64, generator=torch.Generator().manual_seed(1234))
    ge, gt = torch.Generator().manual_seed(1235), torch.Generator().manual_seed(1236)
    losses = []
    for i in range(10):
        eps = torch.randn(4, 3, 64, 64, generator=ge)
        t = torch.randint(1, 1000, (4,), generator=gt)
        if i == 0:
            assert t.tolist() == [753, 878, 655, 810]
        losses.append(float(tr.step(data[4 * i:4 * i + 4].cuda(), None, eps=eps.cuda(), t=t.cuda())))
    torch.cuda.synchronize()
    for got, want in zip(losses, C1_LOSSES):
        assert abs(got - want) <= 1e-4 + 1e-3 * abs(want), (losses, C1_LOSSES)        # north_star: rtol 1e-3 / atol 1e-4
        assert abs(got - want) < 2e-5, (got, want)                                   # what the fp32 path actually holds
    s1 = sum(float(v.double().sum()) for v in net.state_dict().values())
    # VERDICT gate (rtol 1e-3 / atol 1e-4 on the checksum), then the discriminating one: 10 steps move the checksum by
    # +5.0967, and first Adam steps are ~lr*sign(g), so the ~1e4 parameters whose true gradient is a cancelling sum (conv
    # biases ahead of a GroupNorm) random-walk by ~1e-2 between ANY two summation orders; a wrong or stale gradient on a
    # real tensor moves it by >= 0.5
    print(f"C1 KAT: checksum {s1:.6f} (reference {C1_FINAL_SUM:.6f}, diff {s1 - C1_FINAL_SUM:+.2e})")
    assert abs(s1 - C1_FINAL_SUM) <= 1e-4 + 1e-3 * C1_FINAL_SUM
    assert abs((s1 - s0) - (C1_FINAL_SUM - C1_INIT_SUM)) < 5e-2, (s1 - s0, C1_FINAL_SUM - C1_INIT_SUM)


# ------------------------------------------------------------------------------------------------ embedding broadcast
def test_embedding_broadcasts_like_the_reference(dev, O):
    """custom_layers.py:92-98: time_emb [T, D] + cond_emb [n, D] broadcast.  The samplers pass t of shape [1] with labels
    [n, cond_dim] at every checkpoint plot (train_diffusion.py:189-195); one label row with t [N] is the reverse case."""
    from models.custom_layers import ConditionalEmbedding
    cfg = O.full_cfg(dict(time_dim=32, cond_dim=3, num_layers=1, attn_layers=[], min_channel=32, max_channel=32,
                          num_resnet_blocks=1))
    P = O.seeded_params(cfg, seed=5, gain=1.5)
    emb = ConditionalEmbedding(32, 3)
    emb.load_state_dict({k[len("cond_emb."):]: v for k, v in P.items() if k.startswith("cond_emb.")}, strict=True)
    emb = emb.cuda()
    labels = torch.tensor([[1., 0, 0], [0, 1, 0], [0, 0, 1], [.5, .5, 0], [0, 0, 0]])
    for t, c in ((torch.tensor([417]), labels), (torch.tensor([3, 999, 40, 1, 500]), labels[1]),
                 (torch.tensor([3, 999, 40, 1, 500]), labels), (torch.tensor([7]), labels[2]),
                 (torch.tensor([417]), labels.repeat(8, 1))):                 # 40 rows: the matrix-core path (M >= 32)
        ref = O.cond_embedding(P, cfg, t, c)
        out = emb(t.cuda(), c.cuda())
        assert out.shape == ref.shape, (out.shape, ref.shape)
        assert rel(out, ref) < 2e-6, (tuple(t.shape), tuple(c.shape), rel(out, ref))
    with pytest.raises(RuntimeError, match="must match"):
        emb(torch.tensor([1, 2]).cuda(), labels[:3].cuda())
    from sdm_hip import lib, ops
    x = torch.randn(4, 32, device="cuda")
    with pytest.raises(lib.SdmError, match="broadcast"):
        ops.linear_small(x, emb.time_layer[0].weight, emb.time_layer[0].bias, add=torch.zeros(2, 32, device="cuda"))


@pytest.mark.parametrize("use_graph", [False, True])
def test_sampling_with_per_image_labels_vs_oracle(dev, O, use_graph):
    """DDPM and DDIM with labels [n, cond_dim] and the samplers' scalar timestep (the checkpoint-plot call)."""
    import degraders as D
    import diffusion_sampling_algorithms as S
    from models.U_Net import U_Net
    cfg = O.full_cfg(dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3, num_layers=2,
                          attn_layers=[1], num_heads=2, dim_per_head=16, min_channel=32, max_channel=64))
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    net = U_Net(**cfg)
    net.load_state_dict(P, strict=True)
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(8)
    xT = torch.randn(3, 3, 16, 16, generator=g)
    labels = torch.tensor([[1., 0, 0], [0, 0, 1], [0, 1, 0]])
    z = torch.randn(5, 3, 3, 16, 16, generator=g)
    quiet = lambda *a, **k: None
    # min_noise 3: sigma != 0 on the LAST step too, so a mis-indexed final noise tensor shows
    ref, _ = O.ddpm_sample(P, cfg, O.CosineSchedule(1000), xT, z, 3, 7, labels=labels)
    out = S.ddpm_sampling(net, D.CosineNoiseDegradation(1000), xT.cuda(), 3, 7, None, labels.cuda(), "cuda", quiet,
                          noise=z.cuda(), use_graph=use_graph)
    assert rel(out, ref) < 2e-5, rel(out, ref)
    ref, _ = O.ddim_sample(P, cfg, O.CosineSchedule(1000), xT, 1, 1000, 250, labels=labels)
    out = S.ddim_sampling(net, D.CosineNoiseDegradation(1000), xT.cuda(), 1, 1000, None, labels.cuda(), 250, "cuda", quiet,
                          use_graph=use_graph)
    assert rel(out, ref) < 2e-5, rel(out, ref)


# ------------------------------------------------------------------------------------------------ C3
@pytest.mark.parametrize("use_graph", [False, True])
def test_c3_ddim_cfg_bf16_128(dev, O, use_graph):
    """configs[2]: standard architecture, cond_dim=10, 128x128, bf16 storage, 3-step DDIM with guidance scale 3 (one 2N
    batch per step), N=2, against the two-pass fp32 oracle.  The reference has no bf16 mode -> the rounding itself is
    parity-unpinned; the gate is rel-L2 <= 1e-2 (fp32 run of the same sampler: <= 5e-5)."""
    import degraders as D
    import diffusion_sampling_algorithms as S
    cfg, net, P = standard(O, cond_dim=10)
    net.eval()
    g = torch.Generator().manual_seed(77)
    n, w = 2, 3.0
    xT = torch.randn(n, 3, 128, 128, generator=g)
    labels = torch.eye(10)[[4, 9]]
    sched = O.CosineSchedule(1000)
    steps = O.ddim_steps(1, 1000, 400)                                 # [1000, 600, 200, 1]
    def oracle_run(storage):
        x = xT.clone()
        x0 = None
        with torch.no_grad():
            for i, s in enumerate(steps):
                t = torch.full((n,), s)
                e_c = O.unet_forward(P, cfg, x, t, labels, storage=storage)
                e_u = O.unet_forward(P, cfg, x, t, torch.zeros_like(labels), storage=storage)
                e = e_u + w * (e_c - e_u)
                ab = sched.alpha_bar(torch.tensor([s]))
                x0 = (x - (1 - ab) ** 0.5 * e) / ab ** 0.5
                if i + 1 < len(steps):
                    abn = sched.alpha_bar(torch.tensor([steps[i + 1]]))
                    x = abn ** 0.5 * x0 + (1 - abn) ** 0.5 * e
        return x0
    x0 = oracle_run(None)
    quiet = lambda *a, **k: None
    out32 = S.ddim_sampling(net, D.CosineNoiseDegradation(1000), xT.cuda(), 1, 1000, None, labels.cuda(), 400, "cuda", quiet,
                            use_graph=use_graph, cfg_scale=w)
    assert rel(out32, x0) < 5e-5, rel(out32, x0)
    net.set_inference_dtype(torch.bfloat16)
    out16 = S.ddim_sampling(net, D.CosineNoiseDegradation(1000), xT.cuda(), 1, 1000, None, labels.cuda(), 400, "cuda", quiet,
                            use_graph=use_graph, cfg_scale=w)
    assert rel(out16, x0) < 1e-2, rel(out16, x0)
    assert rel(out16, out32) > 1e-6            # the bf16 path really ran
    if use_graph:
        # derived gate (tests/bf16_gates.py; one extra oracle run: the bf16-storage mode of the same guided sampler -- U-Net
        # outputs in storage mode, the sampler state fp32 as in the HIP path; floor = the model's own bf16 error / 4)
        import bf16_gates as BG
        BG.check_forward(out16, oracle_run(torch.bfloat16), x0, None, what="C3 guided DDIM, 4 U-Net passes")


# ------------------------------------------------------------------------------------------------ C4
def test_c4_training_step_128(dev, O):
    """configs[3]: one DDPM training step of the standard architecture at 128x128 (attention backward at S=1024, weight
    gradients over 128*128*N pixels), N=2: loss and all 148 live gradient tensors against the oracle."""
    import degraders as D
    cfg, net, P = standard(O)
    g = torch.Generator().manual_seed(123)
    x0 = torch.randn(2, 3, 128, 128, generator=g)
    eps = torch.randn(2, 3, 128, 128, generator=g)
    t = torch.tensor([901, 88])
    ref_loss, _, ref_g = O.loss_and_grads(P, cfg, O.CosineSchedule(1000), x0, t, eps, None)
    x_t = D.CosineNoiseDegradation(1000)(x0.cuda(), t.cuda(), eps.cuda())
    loss, grads = net.loss_and_grads(x_t, t.cuda(), None, eps.cuda())
    assert abs(float(loss) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
    assert check_grads(O, cfg, net, grads, ref_g) == 148


# ------------------------------------------------------------------------------------------------ C5
def _sr_setup(O, seed=5):
    cfg, net, P = standard(O, in_channel=6, image_recon=True)
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(1, 3, 256, 256, generator=g)
    eps = torch.randn(1, 3, 256, 256, generator=g)
    t = torch.tensor([640])
    return cfg, net, P, x0, eps, t


def _sr_oracle(O, cfg, P, x0, eps, t, cond_t=100):
    x_in, target = O.sr_batch(O.CosineSchedule(1000), x0, t, eps, 64, 256, cond_t)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = O.unet_forward(leaves, cfg, x_in, t, None)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items()}


def test_c5_sr_training_step_256(dev, O):
    """configs[4] in fp32: SR 64->256 training step (train_SR_diffusion.py:318-374: area down/up pair, fixed cond_t noising
    of the low-res image, 6-channel input, Tanh head, residual target), N=1 at 256x256 (attention at S=4096, d=512)."""
    import degraders as D
    from sdm_hip.trainer import DiffusionTrainer
    cfg, net, P, x0, eps, t = _sr_setup(O)
    ref_loss, ref_g = _sr_oracle(O, cfg, P, x0, eps, t)
    tr = DiffusionTrainer(net, D.CosineNoiseDegradation(1000), lr=0.0, sr=(64, 256, 100))      # lr 0: gradients only
    loss = tr.step(x0.cuda(), None, eps=eps.cuda(), t=t.cuda())
    # lr = 0 leaves the weights alone, but the trainer clears each gradient bucket after its Adam: recompute for the check
    x_in, target = tr.sr_inputs(x0.cuda(), eps.cuda(), t.cuda())
    loss2, grads = net.loss_and_grads(x_in, t.cuda(), None, target)
    assert abs(float(loss) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
    assert abs(float(loss2) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
    assert check_grads(O, cfg, net, grads, ref_g) == 148


# ------------------------------------------------------------------------------------------------ C4 / C5 in bf16
def _check_grads_bf16(O, cfg, net, grads, ref_g, lim=2e-2):
    """bf16 training path: per-gradient-tensor rel-L2 <= 2e-2 against the fp32 oracle (the reference has no bf16 mode: the
    rounding itself is parity-unpinned).  Conv biases ahead of a GroupNorm are excluded: their true gradient is a
    cancelling sum ~1e-7 of its terms, below bf16 resolution of the terms."""
    dead = set(O.dead_param_names(cfg))
    bad, errs = [], []
    for k, p in net.named_parameters():
        if k in dead or (".conv_block_" in k and k.endswith("conv_layer.0.bias")):
            continue
        r = rel(grads[p], ref_g[k])
        errs.append((r, k))
        if r > lim:
            bad.append((k, r))
    errs.sort()
    print(f"bf16 gradients vs fp32 oracle: median rel-L2 {errs[len(errs) // 2][0]:.2e}, worst {errs[-1][0]:.2e} ({errs[-1][1]})")
    assert not bad, bad[:10]
    return len(errs)


def test_c4_training_step_128_bf16(dev, O):
    """configs[3] shape on the bf16 training path: loss within 1e-2 relative, gradients within rel-L2 2e-2 of the oracle."""
    import degraders as D
    cfg, net, P = standard(O)
    net.set_training_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(123)
    x0 = torch.randn(2, 3, 128, 128, generator=g)
    eps = torch.randn(2, 3, 128, 128, generator=g)
    t = torch.tensor([901, 88])
    ref_loss, _, ref_g = O.loss_and_grads(P, cfg, O.CosineSchedule(1000), x0, t, eps, None)
    x_t = D.CosineNoiseDegradation(1000)(x0.cuda(), t.cuda(), eps.cuda())
    loss, grads = net.loss_and_grads(x_t, t.cuda(), None, eps.cuda())
    assert abs(float(loss) - float(ref_loss)) < 1e-2 * max(1.0, abs(float(ref_loss)))
    assert _check_grads_bf16(O, cfg, net, grads, ref_g) == 148 - 16
    # derived gate (tests/bf16_gates.py): the oracle's bf16-storage training mode on the same batch (one extra oracle pass)
    import bf16_gates as BG
    x_in = O.q_sample(O.CosineSchedule(1000), x0, t, eps)
    lb, gb = BG.oracle_grads(O, P, cfg, x_in, t, None, eps, O.Bf16Storage(train=True))
    assert abs(float(loss) - lb) <= 3 * max(abs(lb - float(ref_loss)), 5e-5 * abs(float(ref_loss))), (float(loss), lb, float(ref_loss))
    dead = set(O.dead_param_names(cfg))
    hip = {k: grads[p] for k, p in net.named_parameters() if k not in dead}
    BG.check_grads(hip, gb, ref_g, None, skip=lambda k: ".conv_block_" in k and k.endswith("conv_layer.0.bias"), what="C4 128x128 N=2")


def test_c5_sr_training_step_256_bf16(dev, O):
    """configs[4] in the mode BASELINE names it: SR 64->256 training step, in_channel=6, Tanh head, bf16 (attention core at
    S=4096, d=512 and d=1024 through the bf16 kernels, forward and backward)."""
    import degraders as D
    from sdm_hip.trainer import DiffusionTrainer
    cfg, net, P, x0, eps, t = _sr_setup(O)
    ref_loss, ref_g = _sr_oracle(O, cfg, P, x0, eps, t)
    tr = DiffusionTrainer(net, D.CosineNoiseDegradation(1000), lr=0.0, sr=(64, 256, 100), train_dtype=torch.bfloat16)
    x_in, target = tr.sr_inputs(x0.cuda(), eps.cuda(), t.cuda())
    loss, grads = net.loss_and_grads(x_in, t.cuda(), None, target)
    assert abs(float(loss) - float(ref_loss)) < 1e-2 * max(1.0, abs(float(ref_loss)))
    # ONE image: a weight gradient averages over a single sample, and the tensors furthest from the loss (down path, ~40
    # bf16 roundings of the activation gradient behind them) reach 2.7e-2 (median 6e-3); C4 at N=2 holds 2e-2
    assert _check_grads_bf16(O, cfg, net, grads, ref_g, lim=3e-2) == 148 - 16
    # derived gate (tests/bf16_gates.py): the oracle's bf16-storage training mode on the same SR batch (one extra oracle pass)
    import bf16_gates as BG
    xo, to = O.sr_batch(O.CosineSchedule(1000), x0, t, eps, 64, 256, 100)
    lb, gb = BG.oracle_grads(O, P, cfg, xo, t, None, to, O.Bf16Storage(train=True))
    assert abs(float(loss) - lb) <= 3 * max(abs(lb - float(ref_loss)), 5e-5 * abs(float(ref_loss))), (float(loss), lb, float(ref_loss))
    dead = set(O.dead_param_names(cfg))
    hip = {k: grads[p] for k, p in net.named_parameters() if k not in dead}
    BG.check_grads(hip, gb, ref_g, None, skip=lambda k: ".conv_block_" in k and k.endswith("conv_layer.0.bias"), what="C5 SR 256x256 N=1")


# ------------------------------------------------------------------------------------------------ full batch sizes: properties
@pytest.mark.parametrize("which,dtype", [("c4", "f32"), ("c4", "bf16"), ("c5", "f32"), ("c5", "bf16"), ("c3", "bf16")])
def test_full_batch_sizes_through_size_independent_properties(dev, O, which, dtype):
    """BASELINE's per-GPU batch sizes (C3: 64 + 64 guidance images of 128x128; C4: 32 images of 128x128; C5: 8 images of
    256x256, 6-channel SR net) are out of the CPU
    oracle's reach, and the kernels plan differently there (tile sizes, split-K, 256x256 bf16 tiles, cooperative GroupNorm
    tail).  Two properties tie the full-size launches to the small-batch ones the oracle pins above:
      batch invariance -- image i's output does not depend on its batch mates: the batch-N forward equals the forwards of its
        pairs (C4) / single images (C5), row for row;
      gradient linearity -- the mean-loss gradient of the full batch equals the mean of the gradients of its quarters.
    fp32 within summation order (observed 3e-7 / 5e-6; gates 5e-6 / 1e-4); bf16 within bf16 rounding of differently tiled
    sums (observed 9e-4 / 1.2e-2; gates 5e-3 / 3e-2)."""
    n, res, over, sub = {"c4": (32, 128, {}, 2), "c5": (8, 256, dict(in_channel=6, image_recon=True), 1),
                         "c3": (128, 128, dict(cond_dim=10), 2)}[which]     # C3: 64 images + their unconditional twins (CFG)
    cfg, net, _ = standard(O, **over)
    bf = dtype == "bf16"
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, cfg["in_channel"], res, res, generator=g).cuda()
    t = torch.randint(1, 1000, (n,), generator=g).cuda()
    target = torch.randn(n, 3, res, res, generator=g).cuda()
    lab = None
    if which == "c3":
        lab = torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 10).float().cuda()
        lab[n // 2:] = 0.0                                     # the unconditional half of a guidance batch
    net.eval()
    if bf:
        net.set_inference_dtype(torch.bfloat16)
    with torch.no_grad():
        full = net(x, t, lab).float()
        parts = torch.cat([net(x[i:i + sub], t[i:i + sub], None if lab is None else lab[i:i + sub]).float()
                           for i in range(0, n, sub)], dim=0)
    assert bool(torch.isfinite(full).all())
    worst = max(rel(full[i], parts[i]) for i in range(n))
    assert worst < (5e-3 if bf else 5e-6), worst
    print(f"{which} {dtype}: batch invariance worst rel-L2 {worst:.2e}")
    if which == "c3":          # a sampling configuration: no training step to check
        return
    # gradient linearity over quarters of the batch (mean-reduced MSE: the full gradient is the mean of the four)
    net.train()
    if bf:
        net.set_training_dtype(torch.bfloat16)
    _, grads = net.loss_and_grads(x, t, None, target)
    full_g = {p: v.clone() for p, v in grads.items() if v is not None}
    q = n // 4
    acc = {p: torch.zeros_like(v) for p, v in full_g.items()}
    for i in range(0, n, q):
        _, gq = net.loss_and_grads(x[i:i + q], t[i:i + q], None, target[i:i + q])
        for p in acc:
            acc[p] += gq[p] / 4
    dead = set(O.dead_param_names(cfg))
    names = {p: k for k, p in net.named_parameters()}
    bad, worst_g = [], 0.0
    for p, v in full_g.items():
        k = names[p]
        if k in dead or (".conv_block_" in k and k.endswith("conv_layer.0.bias")):      # cancelling sums (see check_grads)
            continue
        r = rel(v, acc[p])
        worst_g = max(worst_g, r)
        if r > (3e-2 if bf else 1e-4):
            bad.append((k, r))
    print(f"{which} {dtype}: gradient linearity worst rel-L2 {worst_g:.2e}")
    assert not bad, bad[:8]
