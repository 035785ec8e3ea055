"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, CPU) on seeded inputs.  Run only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference itself never travels; only the input/output vectors written here do.
Weights are not stored: they are regenerated from ``oracle.unet_oracle.seeded_params``
(numpy PCG64, independent of torch's RNG stream) on both sides.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402

sys.path.insert(0, "/root/reference")
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))  # imported, never used numerically
from models.U_Net import U_Net  # noqa: E402
from models import custom_layers as RL  # noqa: E402
import degraders as RD  # noqa: E402
import diffusion_sampling_algorithms as RS  # noqa: E402

torch.set_num_threads(8)

MODEL_CASES = {
    # name: (cfg, N, H, W, gain)
    "narrow_cond": (dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3,
                         num_layers=2, attn_layers=[1], num_heads=2, dim_per_head=16,
                         min_channel=32, max_channel=64), 3, 16, 24, 1.0),
    "narrow_gain": (dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3,
                         num_layers=2, attn_layers=[1], num_heads=2, dim_per_head=16,
                         min_channel=32, max_channel=64), 2, 16, 16, 1.8),
    "deep_uncond": (dict(num_resnet_blocks=2, in_channel=3, out_channel=3, time_dim=64, cond_dim=None,
                         num_layers=3, attn_layers=[0, 2], num_heads=1, dim_per_head=None,
                         min_channel=32, max_channel=64), 2, 16, 8, 1.5),
    "sr_recon": (dict(num_resnet_blocks=1, in_channel=6, out_channel=3, time_dim=32, cond_dim=None,
                      num_layers=2, attn_layers=[0, 1], num_heads=1, dim_per_head=None,
                      min_channel=32, max_channel=32, image_recon=True), 2, 8, 8, 1.8),
    "no_emb": (dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=None, cond_dim=None,
                    num_layers=2, attn_layers=[1], num_heads=1, dim_per_head=8,
                    min_channel=32, max_channel=64), 2, 8, 12, 1.5),
}


def ref_model(cfg, P):
    m = U_Net(**cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(P.keys()), "oracle.param_shapes order != reference state_dict order"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    m.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    return m


def sample_idx(numel, seed, n=128):
    rng = np.random.default_rng([977, seed])
    return rng.integers(0, numel, size=min(n, numel))


def model_case(name, cfg, N, H, W, gain):
    full = O.full_cfg(cfg)
    P = O.seeded_params(full, seed=3, gain=gain)
    m = ref_model(full, P)
    rng = np.random.default_rng([11, len(name)])
    x = torch.tensor(rng.standard_normal((N, full["in_channel"], H, W)), dtype=torch.float32)
    t = torch.tensor(rng.integers(1, 1000, size=(N,)), dtype=torch.int64)
    cond = None
    if full["cond_dim"] is not None:
        cond = torch.tensor(np.eye(full["cond_dim"])[rng.integers(0, full["cond_dim"], N)], dtype=torch.float32)
    target = torch.tensor(rng.standard_normal((N, full["out_channel"], H, W)), dtype=torch.float32)

    m.eval()
    with torch.no_grad():
        y_eval = m(x, t if full["time_dim"] is not None else None, cond)
        # batch-1 embedding broadcast (sampler call convention, quirk 7)
        if full["time_dim"] is not None:
            t1 = t[:1]
            c1 = cond[0] if cond is not None else None      # shape (cond_dim,) like the generator passes
            y_b1 = m(x, t1, c1)
        else:
            y_b1 = y_eval
    m.train()
    m.zero_grad()
    y = m(x, t if full["time_dim"] is not None else None, cond)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    out = dict(cfg=json.dumps(full), gain=gain, x=x.numpy(), t=t.numpy(), target=target.numpy(),
               y=y_eval.numpy(), y_b1=y_b1.numpy(), loss=float(loss))
    if cond is not None:
        out["cond"] = cond.numpy()
    dead = []
    for i, (k, p) in enumerate(m.named_parameters()):
        if p.grad is None:
            dead.append(k)
            continue
        g = p.grad.detach().double().flatten()
        idx = sample_idx(g.numel(), i)
        out[f"gidx/{k}"] = idx
        out[f"gval/{k}"] = g[idx].float().numpy()
        out[f"gstat/{k}"] = np.array([float(g.sum()), float(g.abs().sum()), float(g.norm())])
    out["dead"] = json.dumps(dead)
    assert dead == O.dead_param_names(full), (dead[:3], O.dead_param_names(full)[:3])
    np.savez(os.path.join(HERE, f"model_{name}.npz"), **out)
    print(name, "params", sum(p.numel() for p in m.parameters()), "loss", float(loss),
          "mean|y|", float(y_eval.abs().mean()))


def block_cases():
    out = {}
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g)

    def put(prefix, mod, **tensors):
        for k, v in mod.state_dict().items():
            out[f"{prefix}/p/{k}"] = v.numpy().copy()
        for k, v in tensors.items():
            out[f"{prefix}/{k}"] = v.detach().numpy().copy()

    def randomise(mod, scale=1.0):
        with torch.no_grad():
            for n_, p in mod.named_parameters():
                if p.dim() == 1:
                    p.copy_(0.3 * r(*p.shape) + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0))
                else:
                    p.copy_(r(*p.shape) * scale / (p[0].numel() ** 0.5))

    # attention: large logits so the softmax axis matters
    for tag, C, heads, dk, H, W in (("attn_h2", 32, 2, 16, 5, 7), ("attn_h1", 32, 1, None, 5, 7)):
        a = RL.AttentionBlock(C, heads=heads, d_k=dk)
        randomise(a)
        with torch.no_grad():
            a.projection.weight.mul_(12.0)
        x = r(2, C, H, W).requires_grad_(True)
        y = a(x)
        dy = r(*y.shape)
        y.backward(dy)
        put(tag, a, x=x, y=y, dy=dy, dx=x.grad,
            dproj_w=a.projection.weight.grad, dproj_b=a.projection.bias.grad,
            dout_w=a.output.weight.grad, dout_b=a.output.bias.grad)

    # AdaGN (emb batch N and batch 1)
    ag = RL.AdaGN(16, 64)
    randomise(ag)
    x = r(3, 64, 6, 10).requires_grad_(True)
    emb = r(3, 16).requires_grad_(True)
    y = ag(x, emb)
    dy = r(*y.shape)
    y.backward(dy)
    put("adagn", ag, x=x, emb=emb, y=y, dy=dy, dx=x.grad, demb=emb.grad,
        dys_w=ag.y_scale.weight.grad, dys_b=ag.y_scale.bias.grad,
        dgn_w=ag.group_norm.weight.grad, dgn_b=ag.group_norm.bias.grad)
    with torch.no_grad():
        out["adagn/y_b1"] = ag(x, emb[:1]).numpy().copy()

    # conv block with embedding, residual block
    cb = RL.UNet_ConvBlock(32, 64, use_activation=True, emb_dim=16)
    randomise(cb)
    x = r(2, 32, 6, 10).requires_grad_(True)
    emb = r(2, 16)
    y = cb(x, emb)
    dy = r(*y.shape)
    y.backward(dy)
    put("convblock", cb, x=x, emb=emb, y=y, dy=dy, dx=x.grad, dw=cb.conv_layer[0].weight.grad,
        db=cb.conv_layer[0].bias.grad)

    rb = RL.ResidualBlock(32, 32, emb_dim=16)
    randomise(rb)
    x = r(2, 32, 6, 10).requires_grad_(True)
    y = rb(x, emb)
    dy = r(*y.shape)
    y.backward(dy)
    put("resblock", rb, x=x, emb=emb, y=y, dy=dy, dx=x.grad)

    dn = RL.DownsampleBlock(32, 64)
    randomise(dn)
    x = r(2, 32, 6, 10).requires_grad_(True)
    y = dn(x)
    dy = r(*y.shape)
    y.backward(dy)
    put("down", dn, x=x, y=y, dy=dy, dx=x.grad, dw=dn.conv_layer[0].weight.grad, db=dn.conv_layer[0].bias.grad)

    up = RL.UpsampleBlock(64, 32)
    randomise(up)
    x = r(2, 64, 3, 5).requires_grad_(True)
    y = up(x)
    dy = r(*y.shape)
    y.backward(dy)
    put("up", up, x=x, y=y, dy=dy, dx=x.grad, dw=up.conv_layer[0].weight.grad, db=up.conv_layer[0].bias.grad)

    for tag, cd in (("emb_t", None), ("emb_tc", 5)):
        ce = RL.ConditionalEmbedding(32, cd)
        randomise(ce)
        t = torch.tensor([1, 500, 1000])
        cond = r(3, cd) if cd else None
        y = ce(t, cond)
        kw = dict(t=t, y=y)
        if cd:
            kw["cond"] = cond
        put(tag, ce, **kw)
    np.savez(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks:", len(out), "arrays")


def schedule_and_sampler_cases():
    out = {}
    ts = torch.tensor([1, 2, 500, 999, 1000])
    lin = RD.NoiseDegradation(5e-3, 9e-3, 1000)
    cos = RD.CosineNoiseDegradation(1000)
    for tag, s in (("lin", lin), ("cos", cos)):
        b, a, ab = s.get_timestep_params(ts)
        out[f"sched/{tag}"] = torch.stack([b, a, ab]).numpy()
    out["sched/ts"] = ts.numpy()
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(3, 3, 8, 8, generator=g)
    eps = torch.randn(3, 3, 8, 8, generator=g)
    tq = torch.tensor([1, 400, 1000])
    out["q/x0"], out["q/eps"], out["q/t"] = x0.numpy(), eps.numpy(), tq.numpy()
    out["q/lin"] = lin(x0, tq, eps).numpy()
    out["q/cos"] = cos(x0, tq, eps).numpy()

    cfg = O.full_cfg(MODEL_CASES["narrow_cond"][0])
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    m = ref_model(cfg, P)
    xT = torch.randn(2, 3, 8, 8, generator=g)
    labels = torch.tensor([0.0, 1.0, 0.0])            # shape (cond_dim,), as generate_images_diffusion.py:183
    quiet = lambda *a, **k: None
    out["samp/xT"], out["samp/labels"] = xT.numpy(), labels.numpy()

    for tag, sch in (("lin", lin), ("cos", cos)):
        # DDPM over steps 6..1 ; z drawn by torch.randn(shape) after each forward
        torch.manual_seed(77)
        x_end = RS.ddpm_sampling(m, sch, xT.clone(), min_noise=1, max_noise=6, labels_tensor=labels, log=quiet)
        torch.manual_seed(77)
        zs = torch.stack([torch.randn(xT.shape) for _ in range(5)])
        out[f"samp/ddpm_{tag}_z"] = zs.numpy()
        out[f"samp/ddpm_{tag}_out"] = x_end.numpy()
    # DDIM 1000 -> 1 in steps of 250 (+[1]) : [1000, 750, 500, 250, 1]
    for tag, sch in (("lin", lin), ("cos", cos)):
        torch.manual_seed(78)
        x_end = RS.ddim_sampling(m, sch, xT.clone(), min_noise=1, max_noise=1000, labels_tensor=labels,
                                 ddim_step_size=250, log=quiet)
        out[f"samp/ddim_{tag}_out"] = x_end.numpy()
    # DDIM whose last step is not 1 (returns x_t, not x0): min_noise=3
    torch.manual_seed(79)
    x_end = RS.ddim_sampling(m, lin, xT.clone(), min_noise=3, max_noise=13, labels_tensor=labels,
                             ddim_step_size=5, log=quiet)
    out["samp/ddim_min3_out"] = x_end.numpy()
    np.savez(os.path.join(HERE, "schedules_samplers.npz"), **out)
    print("schedules/samplers:", len(out), "arrays")


def train_case():
    """3 Adam steps of the reference training-step body (train_diffusion.py:307-366) on the narrow net."""
    cfg = O.full_cfg(MODEL_CASES["narrow_cond"][0])
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    m = ref_model(cfg, P)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, betas=(0.5, 0.999))
    cos = RD.CosineNoiseDegradation(1000)
    g = torch.Generator().manual_seed(21)
    out = {}
    losses = []
    for i in range(3):
        x0 = torch.randn(4, 3, 16, 16, generator=g)
        eps = torch.randn(4, 3, 16, 16, generator=g)
        t = torch.randint(1, 1000, (4,), generator=g)
        lab = torch.eye(3)[torch.randint(0, 3, (4,), generator=g)]
        opt.zero_grad()
        x_t = cos(x0, t, eps)
        loss = torch.nn.functional.mse_loss(m(x_t, t, lab), eps)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        out[f"x0_{i}"], out[f"eps_{i}"], out[f"t_{i}"], out[f"lab_{i}"] = x0.numpy(), eps.numpy(), t.numpy(), lab.numpy()
    out["losses"] = np.array(losses)
    for i, (k, v) in enumerate(m.state_dict().items()):
        v = v.double().flatten()
        idx = sample_idx(v.numel(), i)
        out[f"pidx/{k}"] = idx
        out[f"pval/{k}"] = v[idx].float().numpy()
        out[f"pstat/{k}"] = np.array([float(v.sum()), float(v.abs().sum())])
    np.savez(os.path.join(HERE, "train_narrow.npz"), **out)
    print("train losses", losses)


def sr_case():
    """Super-resolution path of the reference: two Adam steps of the train_SR_diffusion.py step body (:318-374) and a
    cold_diffusion_sampling run (diffusion_sampling_algorithms.py:150-217) with a conditional low-res image."""
    import torch.nn.functional as F
    cfg = O.full_cfg(MODEL_CASES["sr_recon"][0])
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    m = ref_model(cfg, P)
    lin = RD.NoiseDegradation(5e-3, 9e-3, 1000)
    g = torch.Generator().manual_seed(31)
    out = {}
    # sampler first (fresh weights)
    lr_img = torch.randn(2, 3, 4, 4, generator=g)
    noise = torch.randn(2, 3, 8, 8, generator=g)
    up = F.interpolate(lr_img, size=(8, 8), mode="area")
    cond = lin(img=up, steps=torch.tensor([250]), eps=noise)
    m.eval()
    quiet = lambda *a, **k: None
    x0 = RS.cold_diffusion_sampling(m, lin, 1 * noise, noise, min_noise=1, max_noise=1000, cond_img=cond,
                                    skip_step_size=250, log=quiet)
    out["samp/lr_img"], out["samp/noise"], out["samp/up"], out["samp/cond"], out["samp/x0"] = (
        lr_img.numpy(), noise.numpy(), up.numpy(), cond.numpy(), x0.detach().numpy())
    # area interpolation goldens (down by 2, up by 2, non-integer ratio)
    a = torch.randn(2, 3, 12, 8, generator=g)
    out["area/x"] = a.numpy()
    for tag, size in (("down", (6, 4)), ("up", (24, 16)), ("odd", (5, 3)), ("mixed", (9, 12))):
        out[f"area/{tag}"] = F.interpolate(a, size=size, mode="area").numpy()
    # training
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, betas=(0.5, 0.999))
    losses = []
    for i in range(2):
        tr = torch.randn(4, 3, 8, 8, generator=g)
        eps = torch.randn(4, 3, 8, 8, generator=g)
        t = torch.randint(1, 1000, (4,), generator=g)
        lr = F.interpolate(F.interpolate(tr, size=(4, 4), mode="area"), size=(8, 8), mode="area")
        opt.zero_grad()
        x_in = torch.cat((lin(tr, t, eps), lin(lr, torch.tensor([250]), eps)), dim=1)
        loss = F.mse_loss(m(x_in, t, None), tr - lr)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        out[f"train/x0_{i}"], out[f"train/eps_{i}"], out[f"train/t_{i}"] = tr.numpy(), eps.numpy(), t.numpy()
    out["train/losses"] = np.array(losses)
    for i, (k, v) in enumerate(m.state_dict().items()):
        v = v.double().flatten()
        idx = sample_idx(v.numel(), i)
        out[f"pidx/{k}"] = idx
        out[f"pval/{k}"] = v[idx].float().numpy()
    np.savez(os.path.join(HERE, "sr_path.npz"), **out)
    print("sr losses", losses)


def _params_sample(m, out):
    for i, (k, v) in enumerate(m.state_dict().items()):
        v = v.double().flatten()
        idx = sample_idx(v.numel(), i)
        out[f"pidx/{k}"] = idx
        out[f"pval/{k}"] = v[idx].float().numpy()


COLD_CFG = dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3, num_layers=2, attn_layers=[1],
                num_heads=2, dim_per_head=16, min_channel=32, max_channel=64, image_recon=True)
DOODLE_CFG = dict(num_resnet_blocks=1, in_channel=6, out_channel=3, time_dim=32, cond_dim=None, num_layers=2,
                  attn_layers=[0], num_heads=1, dim_per_head=None, min_channel=32, max_channel=64)


def cold_and_doodle_cases():
    """Two Adam steps each of the cold-diffusion step body (train_noise_cold_diffusion.py:307-348: x0 target, labels,
    Tanh head) and of the doodle step body (train_doodle_diffusion.py:283-320: cat((x_t, lbl_data), 1), noise target), plus
    a doodle DDIM run with the label image as ``cond_img`` (train_doodle_diffusion.py:398-410)."""
    F = torch.nn.functional
    cos = RD.CosineNoiseDegradation(1000)
    g = torch.Generator().manual_seed(41)
    out = {"cold/cfg": json.dumps(COLD_CFG), "doodle/cfg": json.dumps(DOODLE_CFG)}
    # ---- cold
    cfg = O.full_cfg(COLD_CFG)
    m = ref_model(cfg, O.seeded_params(cfg, seed=3, gain=1.8))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, betas=(0.5, 0.999))
    losses = []
    for i in range(2):
        x0 = torch.tanh(torch.randn(4, 3, 16, 16, generator=g))
        eps = torch.randn(4, 3, 16, 16, generator=g)
        t = torch.randint(1, 1000, (4,), generator=g)
        lab = torch.eye(3)[torch.randint(0, 3, (4,), generator=g)]
        opt.zero_grad()
        loss = F.mse_loss(m(cos(x0, t, eps), t, lab), x0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        out[f"cold/x0_{i}"], out[f"cold/eps_{i}"], out[f"cold/t_{i}"], out[f"cold/lab_{i}"] = (
            x0.numpy(), eps.numpy(), t.numpy(), lab.numpy())
    out["cold/losses"] = np.array(losses)
    sub = {}
    _params_sample(m, sub)
    out.update({"cold/" + k: v for k, v in sub.items()})
    print("cold losses", losses)
    # ---- doodle
    cfg = O.full_cfg(DOODLE_CFG)
    m = ref_model(cfg, O.seeded_params(cfg, seed=3, gain=1.8))
    m.eval()
    quiet = lambda *a, **k: None
    xT = torch.randn(2, 3, 16, 16, generator=g)
    lbl = torch.tanh(torch.randn(2, 3, 16, 16, generator=g))
    x_end = RS.ddim_sampling(m, cos, 1 * xT, min_noise=1, max_noise=1000, cond_img=lbl, labels_tensor=None,
                             ddim_step_size=250, log=quiet)
    out["doodle/samp_xT"], out["doodle/samp_lbl"], out["doodle/samp_out"] = xT.numpy(), lbl.numpy(), x_end.detach().numpy()
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, betas=(0.5, 0.999))
    losses = []
    for i in range(2):
        x0 = torch.randn(4, 3, 16, 16, generator=g)
        lb = torch.tanh(torch.randn(4, 3, 16, 16, generator=g))
        eps = torch.randn(4, 3, 16, 16, generator=g)
        t = torch.randint(1, 1000, (4,), generator=g)
        opt.zero_grad()
        loss = F.mse_loss(m(torch.cat((cos(x0, t, eps), lb), dim=1), t, None), eps)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        out[f"doodle/x0_{i}"], out[f"doodle/lbl_{i}"], out[f"doodle/eps_{i}"], out[f"doodle/t_{i}"] = (
            x0.numpy(), lb.numpy(), eps.numpy(), t.numpy())
    out["doodle/losses"] = np.array(losses)
    sub = {}
    _params_sample(m, sub)
    out.update({"doodle/" + k: v for k, v in sub.items()})
    print("doodle losses", losses)
    np.savez(os.path.join(HERE, "cold_doodle.npz"), **out)


CKPT_CFG = dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=2, num_layers=1, attn_layers=[0],
                num_heads=1, dim_per_head=None, min_channel=32, max_channel=32)


def checkpoint_case():
    """A checkpoint written the way the REAL reference writes it -- torch.save({"model": state_dict, "optimizer":
    Adam.state_dict()}) (train_diffusion.py:384-395, utils/utils.py:67-95) after one optimiser step of the reference
    model -- a config checkpoint ({"starting_epoch", "global_steps"}, train_diffusion.py:375-383), and the inference
    config.json that export_models.py:72-99 derives from a training config.  Plus what the reference produces from it:
    the forward on a fixed input and the parameters after ONE more Adam step resumed from the saved optimizer state."""
    cfg = O.full_cfg(CKPT_CFG)
    m = ref_model(cfg, O.seeded_params(cfg, seed=7, gain=1.8))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, betas=(0.5, 0.999))
    cos = RD.CosineNoiseDegradation(1000)
    g = torch.Generator().manual_seed(51)
    F = torch.nn.functional

    def step():
        x0 = torch.randn(4, 3, 8, 8, generator=g)
        eps = torch.randn(4, 3, 8, 8, generator=g)
        t = torch.randint(1, 1000, (4,), generator=g)
        lab = torch.eye(2)[torch.randint(0, 2, (4,), generator=g)]
        opt.zero_grad()
        loss = F.mse_loss(m(cos(x0, t, eps), t, lab), eps)
        loss.backward()
        opt.step()
        return x0, eps, t, lab, float(loss)

    step()
    d = os.path.join(HERE, "ref_checkpoint")
    os.makedirs(d, exist_ok=True)
    torch.save({"model": m.state_dict(), "optimizer": opt.state_dict()}, os.path.join(d, "diffusion_1.pt"))
    torch.save({"starting_epoch": 0, "global_steps": 1}, os.path.join(d, "config_1.pt"))
    train_cfg = dict(in_channel=3, out_channel=3, num_layers=1, num_resnet_block=1, attn_layers=[0], attn_heads=1,
                     attn_dim_per_head=None, time_dim=32, cond_dim=2, min_channel=32, max_channel=32, img_recon=False,
                     min_noise_step=1, max_noise_step=1000, noise_scheduler="COSINE")
    entry = {"model_name": "diffusion_1.pt", "img_C": 3, "img_H": 8, "img_W": 8}                  # export_models.py:72-92
    entry.update({k: train_cfg[k] for k in ("in_channel", "out_channel", "num_layers", "num_resnet_block", "attn_layers",
                                            "attn_heads", "attn_dim_per_head", "time_dim", "cond_dim", "min_channel",
                                            "max_channel")})
    entry.update(image_recon=train_cfg["img_recon"], max_noise=train_cfg["max_noise_step"],
                 min_noise=train_cfg["min_noise_step"], noise_scheduler=train_cfg["noise_scheduler"])
    # export_models.py:94-97: a BASE model's entry ALWAYS carries beta_1 / beta_T (copied from the training config's
    # "beta1" / "betaT", create_diffusion_config.py:113-122 defaults), whatever the scheduler
    entry.update(beta_1=5e-3, beta_T=9e-3)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"models": [entry]}, f)
    with open(os.path.join(d, "config_linear.json"), "w") as f:       # the same checkpoint exported with the LINEAR schedule
        json.dump({"models": [dict(entry, noise_scheduler="LINEAR")]}, f)
    out = {}
    m.eval()
    x = torch.randn(2, 3, 8, 8, generator=g)
    with torch.no_grad():
        out["fwd_x"], out["fwd_t"], out["fwd_lab"] = x.numpy(), np.array([400, 20]), np.eye(2, dtype=np.float32)
        out["fwd_y"] = m(x, torch.tensor([400, 20]), torch.eye(2)).numpy()
    m.train()
    x0, eps, t, lab, loss = step()
    out["x0"], out["eps"], out["t"], out["lab"], out["loss"] = x0.numpy(), eps.numpy(), t.numpy(), lab.numpy(), np.array(loss)
    _params_sample(m, out)
    # what the reference generator does with config_linear.json (generate_images_diffusion.py:186-192, :232-254):
    # NoiseDegradation(beta_1, beta_T, max_T) + DDIM from a fixed x_T with one label row broadcast over the batch.
    # Uses the model as saved in diffusion_1.pt (one Adam step), drawn AFTER everything above: earlier vectors unchanged.
    m2 = ref_model(cfg, {k: v.clone() for k, v in torch.load(os.path.join(d, "diffusion_1.pt"), weights_only=True)["model"].items()})
    m2.eval()
    lin = RD.NoiseDegradation(entry["beta_1"], entry["beta_T"], 1000, "cpu")
    xT = torch.randn(2, 3, 8, 8, generator=g)
    quiet = lambda *a, **k: None
    out["lin_xT"] = xT.numpy()
    out["lin_ddim_out"] = RS.ddim_sampling(m2, lin, 1 * xT, min_noise=1, max_noise=1000, cond_img=None,
                                           labels_tensor=torch.tensor([1.0, 0.0]), ddim_step_size=250, log=quiet).detach().numpy()
    np.savez(os.path.join(HERE, "ref_checkpoint", "expected.npz"), **out)
    print("checkpoint case: resumed-step loss", loss)


def dataset_case():
    """SURVEY §8f rank 2 pin: the reference's OWN dataset classes (custom_dataset/img_dataset.py:19-36,
    conditional_img_dataset.py:14-66, doodle_dataset.py:14-66) run here with `cv2` and `tinydb` stubbed -- neither is
    installed, and neither does arithmetic: `cv2.imread` returns fixed uint8 BGR arrays (what a decoder hands over), the
    TinyDB stub returns the tables' records in document order.  The normalisation, dtype conversions, permutation, label
    handling and the initial shuffle are the reference's code.  Stored: the decoded bytes (inputs) and the tensors the
    reference's __getitem__ returned (outputs)."""
    import random
    rng = np.random.default_rng(20240301)
    n, H, W = 6, 8, 32
    bgr = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    bgr[0] = (np.arange(H * W * 3) % 256).astype(np.uint8).reshape(H, W, 3)          # every byte value 0..255 occurs
    lab = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    files = {f"/virtual/im{i}.png": bgr[i] for i in range(n)}
    files.update({f"/virtual/lab{i}.png": lab[i] for i in range(n)})
    cv2 = types.ModuleType("cv2")
    cv2.imread = lambda path: files[path].copy()
    recs = [{"filename": f"/virtual/im{i}.png", "doodle": f"/virtual/lab{i}.png", "smiling": i % 2, "hat": 0.25 * i}
            for i in range(n)]
    dbs = {"cond.json": {"Data": recs, "Labels": [{"labels": ["smiling", "hat"]}]},
           "doodle.json": {"Data": recs, "Labels": [{"labels": ["doodle"]}]}}

    class _Table(list):
        def all(self):
            return list(self)

    class TinyDB:
        def __init__(self, path):
            self._db = dbs[os.path.basename(path)]

        def table(self, name):
            return _Table(dict(r) for r in self._db.get(name, []))

    tinydb = types.ModuleType("tinydb")
    tinydb.TinyDB = TinyDB
    sys.modules["cv2"], sys.modules["tinydb"] = cv2, tinydb
    try:
        from custom_dataset.img_dataset import ImageDataset
        from custom_dataset.conditional_img_dataset import ConditionalImgDataset
        from custom_dataset.doodle_dataset import DoodleImgDataset
        out = {"bgr_u8": bgr, "lab_u8": lab}
        ds = ImageDataset([f"/virtual/im{i}.png" for i in range(n)])
        out["img_out"] = np.stack([ds[i].numpy() for i in range(n)])
        assert out["img_out"].dtype == np.float32
        random.seed(123)
        dc = ConditionalImgDataset("cond.json")
        out["cond_order"] = np.array([int(p[len("/virtual/im"):-4]) for p, _ in dc.dataset])
        items = [dc[i] for i in range(n)]
        out["cond_img_out"] = np.stack([a.numpy() for a, _ in items])
        out["cond_labels_out"] = np.stack([b.numpy() for _, b in items])
        assert dc.get_labels() == ["smiling", "hat"]
        random.seed(321)
        dd = DoodleImgDataset("doodle.json")
        out["doodle_order"] = np.array([int(p[len("/virtual/im"):-4]) for p, _ in dd.dataset])
        items = [dd[i] for i in range(n)]
        out["doodle_img_out"] = np.stack([a.numpy() for a, _ in items])
        out["doodle_lab_out"] = np.stack([b.numpy() for _, b in items])
        out["smiling"] = np.array([r["smiling"] for r in recs], dtype=np.float64)
        out["hat"] = np.array([r["hat"] for r in recs], dtype=np.float64)
    finally:
        del sys.modules["cv2"], sys.modules["tinydb"]
    np.savez(os.path.join(HERE, "datasets.npz"), **out)
    print("dataset case: orders", out["cond_order"], out["doodle_order"])


if __name__ == "__main__":
    which = sys.argv[1:]
    if not which:
        for name, (cfg, N, H, W, gain) in MODEL_CASES.items():
            model_case(name, cfg, N, H, W, gain)
        block_cases()
        schedule_and_sampler_cases()
        train_case()
        sr_case()
    if not which or "cold_doodle" in which:
        cold_and_doodle_cases()
    if not which or "checkpoint" in which:
        checkpoint_case()
    if not which or "datasets" in which:
        dataset_case()
