"""CPU oracle for the U-Net denoise hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, *functional* restatement (plain torch CPU ops over a
flat ``{name: tensor}`` parameter dict) of the reference algorithm.  It is the
checker for the HIP path and the ``cpu_baseline`` leg of ``bench.py``; nothing in
the product package (``simple-diffusion-model_amd/``) may import it.

Pinning: ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference/models``) in the build container and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
below against those vectors (parity is therefore *pinned*, not "unpinned").

Reference citations (file:line under /root/reference):
  channel plan ............ models/U_Net.py:41-46
  forward assembly ........ models/U_Net.py:147-173
  sinusoid + MLPs ......... models/custom_layers.py:82-98 (layers :59-78)
  conv block .............. models/custom_layers.py:240-245 (conv -> SiLU -> AdaGN)
  AdaGN (shift==scale) .... models/custom_layers.py:35-45 (note :41 uses y_scale)
  residual block .......... models/custom_layers.py:282-287
  attention ............... models/custom_layers.py:127-163 (softmax(dim=1) = over queries, :147;
                            self.norm never applied)
  down / up sample ........ models/custom_layers.py:204-207 / 182-185
  schedules ............... degraders.py:26-59 (linear), 70-104 (cosine)
  DDPM / DDIM samplers .... diffusion_sampling_algorithms.py:5-64 / 66-148
  training step ........... train_diffusion.py:307-366
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# architecture bookkeeping
# --------------------------------------------------------------------------- #
DEFAULT_CFG = dict(
    num_resnet_blocks=5, in_channel=3, out_channel=3, time_dim=64, cond_dim=None,
    num_layers=5, attn_layers=[2, 3, 4], num_heads=1, dim_per_head=None, groups=32,
    min_channel=128, max_channel=512, image_recon=False)

STANDARD_CFG = dict(
    num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=512, cond_dim=None,
    num_layers=4, attn_layers=[2, 3], num_heads=1, dim_per_head=None, groups=32,
    min_channel=128, max_channel=512, image_recon=False)


def full_cfg(cfg):
    out = dict(DEFAULT_CFG)
    out.update(cfg)
    return out


def channel_plan(cfg):
    """models/U_Net.py:41-46 — doubles until max_channel."""
    ch = [cfg["min_channel"]]
    c = cfg["min_channel"]
    for _ in range(cfg["num_layers"]):
        c *= 2
        ch.append(min(c, cfg["max_channel"]))
    return ch


def param_shapes(cfg):
    """Ordered {state_dict name: shape} exactly as the reference registers them."""
    cfg = full_cfg(cfg)
    ch = channel_plan(cfg)
    D, cd = cfg["time_dim"], cfg["cond_dim"]
    L, R = cfg["num_layers"], cfg["num_resnet_blocks"]
    h = cfg["num_heads"]
    shapes = {}

    def lin(name, i, o):
        shapes[name + ".weight"] = (o, i)
        shapes[name + ".bias"] = (o,)

    def conv(name, i, o, k=3):
        shapes[name + ".weight"] = (o, i, k, k)
        shapes[name + ".bias"] = (o,)

    def convblock(name, i, o, emb):
        conv(name + ".conv_layer.0", i, o)
        if emb:
            lin(name + ".adagn.y_scale", D, o)
            lin(name + ".adagn.y_shift", D, o)
            shapes[name + ".adagn.group_norm.weight"] = (o,)
            shapes[name + ".adagn.group_norm.bias"] = (o,)

    def unetblock(name, cin, cout, use_attn, up):
        for r in range(R):
            convblock(f"{name}.res_layers.{r}.conv_block_1", cin, cin, D is not None)
            convblock(f"{name}.res_layers.{r}.conv_block_2", cin, cin, D is not None)
        for r in range(R):      # ModuleList registration order: all res_layers, then all attn_layers
            if use_attn:
                dk = cfg["dim_per_head"] or cin
                shapes[f"{name}.attn_layers.{r}.norm.weight"] = (cin,)
                shapes[f"{name}.attn_layers.{r}.norm.bias"] = (cin,)
                lin(f"{name}.attn_layers.{r}.projection", cin, 3 * h * dk)
                lin(f"{name}.attn_layers.{r}.output", h * dk, cin)
        if up:
            shapes[f"{name}.out_layer.conv_layer.0.weight"] = (cin, cout, 4, 4)
            shapes[f"{name}.out_layer.conv_layer.0.bias"] = (cout,)
        else:
            conv(f"{name}.out_layer.conv_layer.0", cin, cout)

    if D is not None:
        for j, i in enumerate((0, 2, 4, 6)):
            lin(f"cond_emb.time_layer.{i}", D, D)
        if cd is not None:
            for j, i in enumerate((0, 2, 4, 6)):
                lin(f"cond_emb.cond_layer.{i}", cd if j == 0 else D, D)
    convblock("in_layer.0", cfg["in_channel"], ch[0], False)
    convblock("in_layer.1", ch[0], ch[0], False)
    for k in range(L):
        unetblock(f"down_layers.{k}", ch[k], ch[k + 1], k in cfg["attn_layers"], False)
    convblock("middle_layer.0", ch[-1], ch[-1], False)
    convblock("middle_layer.1", ch[-1], ch[-1], False)
    for idx, k in enumerate(range(L - 1, -1, -1)):
        unetblock(f"up_layers.{idx}", 2 * ch[k + 1], ch[k], k in cfg["attn_layers"], True)
    convblock("out_layers.0", ch[0], ch[0], False)
    convblock("out_layers.1", ch[0], cfg["out_channel"], False)
    return shapes


def dead_param_names(cfg):
    """Parameters that never receive a gradient."""
    return [n for n in param_shapes(cfg)
            if ".adagn.y_shift." in n or (".attn_layers." in n and ".norm." in n)]


def seeded_params(cfg, seed=0, gain=1.0, dtype=torch.float32):
    """Deterministic parameters that do not depend on torch's RNG stream:
    numpy PCG64 per tensor, fan-in-scaled uniform (same family as torch's default
    initialisers), GroupNorm affine jittered so gamma/beta matter."""
    import numpy as np
    params = {}
    for i, (name, shp) in enumerate(param_shapes(cfg).items()):
        rng = np.random.default_rng([seed, i])
        if "group_norm" in name or ".norm." in name:
            base = 1.0 if name.endswith("weight") else 0.0
            a = base + 0.2 * rng.standard_normal(shp)
        else:
            if len(shp) == 4 and ".out_layer." in name and shp[2] == 4:
                fan_in = shp[1] * shp[2] * shp[3]   # ConvTranspose2d: torch uses size(1)*k*k
            elif len(shp) >= 2:
                fan_in = int(np.prod(shp[1:]))
            else:
                fan_in = None
            if fan_in is None:
                a = rng.uniform(-0.1, 0.1, shp)
            else:
                b = gain / math.sqrt(fan_in)
                a = rng.uniform(-b, b, shp)
        params[name] = torch.tensor(a, dtype=dtype)
    return params


# --------------------------------------------------------------------------- #
# bf16 STORAGE mode (round 4): the same formulas with a round-to-nearest-even to bf16 exactly where the HIP path STORES bf16
# and fp32 everywhere it keeps fp32.  The reference has no bf16 mode (its reduced-precision path is fp16 autocast,
# train_diffusion.py:130,333, and holds no fixture), so this is not a pin of the rounding against the reference: it is
# the reference's arithmetic (pinned above, storage=None reproduces every golden bit for bit) composed with a storage
# model that is read off the HIP path's kernel boundaries:
#   forward   conv / convT / token-linear weights: bf16 copies (sdm_cast_bf16; stem conv: fp32 weights and fp32 image);
#             every contraction epilogue: fp32 accumulator -> alpha, bias -> activation -> (+ residual) -> bf16 store;
#             GroupNorm statistics from the STORED (rounded) values, fp32 / fp64 sums; the folded AdaGN affine is applied
#             by sdm_affine_residual_bf16 / the fused tails -> bf16 store; attention: qkv bf16, scores fp32, probabilities
#             bf16 (not rounded at all in the short-sequence no-grad core, which never writes them), P.V -> bf16, output
#             projection + residual -> bf16; head conv: bf16 weights, fp32 result; embedding MLPs, y_scale, samplers: fp32.
#   backward  (training, sdm_bf16_train.hip) activation gradients are stored bf16 at: every conv pre-activation (the
#             data-gradient epilogue multiplies by silu'(z) -- z being the STORED bf16 pre-activation -- adds the skip
#             gradients and rounds once), the materialised AdaGN output, block outputs, qkv / attention result, dS of the
#             column softmax, the two halves of a skip concat, the loss gradient entering the head conv; weight gradients,
#             GroupNorm / modulation / embedding gradients are fp32 sums over those bf16 operands.
# With the storage points equal, what is left between the HIP path and this mode is fp32 evaluation order in front of the
# stores -- and that does NOT stay small through a network: see Bf16Storage (jitter / floor_twin) for why, and for the
# derived gates (a multiple of the model's own flip-noise floor; the fp32-oracle gates stay as the outer bound).  One
# storage stage at a time the agreement is at flip level: a net without GroupNorm / AdaGN (golden case no_emb), whose
# stores are all fed by EXACT fp32 sums of bf16 products, reproduces this mode to 7e-8.
# --------------------------------------------------------------------------- #
def round_bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _store(x, q):
    """fp32 value -> what a bf16 store keeps.  q.jitter > 0 first moves the value by a relative jitter * U(-1, 1): a model
    of fp32 evaluation-order noise (a few 2^-24) in FRONT of the rounding (see Bf16Storage.floor_twin)."""
    if q is not None and q.jitter > 0.0:
        x = x * (1.0 + q.jitter * (2.0 * torch.rand(x.shape, generator=q.gen) - 1.0))
    return round_bf16(x)


class _RoundFwd(torch.autograd.Function):
    """bf16 store of a value whose gradient stays fp32 (weight copies: dW is an fp32 sum; probabilities: dP is fp32)."""
    @staticmethod
    def forward(ctx, x, q):
        return _store(x, q)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBoth(torch.autograd.Function):
    """bf16 store of an activation AND of its gradient (a tensor that exists in HBM in both passes)."""
    @staticmethod
    def forward(ctx, x, q):
        ctx.q = q
        return _store(x, q)

    @staticmethod
    def backward(ctx, g):
        return _store(g, ctx.q), None


class _RoundGrad(torch.autograd.Function):
    """fp32 forward value whose gradient is stored bf16 (scores: dS; a skip half entering the concat; the head's input gradient)."""
    @staticmethod
    def forward(ctx, x, q):
        ctx.q = q
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _store(g, ctx.q), None


class _ActStore(torch.autograd.Function):
    """Contraction epilogue with SiLU on bf16 rows: y = bf16(silu(z)) from the fp32 z; the backward multiplies by silu' of the
    STORED bf16 pre-activation and stores the result (the producer's dz) in bf16."""
    @staticmethod
    def forward(ctx, z, q):
        ctx.q = q
        if q is not None and q.jitter > 0.0:
            z = z * (1.0 + q.jitter * (2.0 * torch.rand(z.shape, generator=q.gen) - 1.0))
        ctx.save_for_backward(round_bf16(z))
        return round_bf16(z * torch.sigmoid(z))

    @staticmethod
    def backward(ctx, g):
        (zb,) = ctx.saved_tensors
        sg = torch.sigmoid(zb)
        return _store(g * (sg * (1 + zb * (1 - sg))), ctx.q), None


class _StemConv(torch.autograd.Function):
    """in_layer.0 on the bf16 training path: fp32 image x fp32 weights forward (sdm_stem_conv_bf16); the weight gradient
    contracts the bf16-rounded image rows (sdm_nchw_to_rows_bf16) with the bf16 dz."""
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.conv2d(x, w, b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gw = torch.nn.grad.conv2d_weight(round_bf16(x), w.shape, g, padding=1)
        return None, gw, g.sum(dim=(0, 2, 3))


class Bf16Storage:
    """Storage model of the HIP bf16 paths (see the block comment above).  ``small_attn(n, heads, S, dk, train)`` -> True
    where the HIP path runs its short-sequence attention core: in a no-grad forward that core never writes the
    probabilities (no rounding of P).  The default mirrors sdm_hip/ops.py:_small_attn_ok (S <= 256, n*heads*S^2 <= 160k
    score elements, and with P kept -- training -- S <= 64); tests assert that the constants agree.

    jitter / seed: the FLOOR of any comparison with this model.  A bf16 store is a discontinuous function of the fp32 value
    in front of it, and two correct implementations differ in that value by fp32 evaluation-order noise (GroupNorm
    statistics, the folded affine, softmax sums: a few 2^-24 relative).  Where the value sits that close to a rounding
    boundary the stored element differs by a whole bf16 ulp (2^-8), the next layer sees a perturbation ~1e-4 and flips
    further roundings: after a handful of layers the two computations carry INDEPENDENT rounding noise (measured on the
    golden nets: a jitter of 2^-24 already moves the output by half of the whole bf16-vs-fp32 distance, and the distance
    saturates there: it does not shrink with the jitter).  So the distance of a correct bf16 implementation to this
    model is not "summation-order noise x rounding count" ~ 1e-4; it is the distance between this model and its own
    jittered twin, which the oracle can compute: floor_twin().  The tests gate the HIP path at a small multiple of that
    floor, and two-sided on the size of its bf16 error (HIP vs fp32 against model vs fp32); single storage stages, where
    nothing can cascade, are gated at flip level (tests/test_gpu_bf16.py)."""

    def __init__(self, train=False, max_s=256, max_work=160 * 1024, keep_p_max_s=64, jitter=0.0, seed=0):
        self.train, self.max_s, self.max_work, self.keep_p_max_s = train, max_s, max_work, keep_p_max_s
        self.jitter = float(jitter)
        self.gen = torch.Generator().manual_seed(seed)

    def floor_twin(self, seed=1, jitter=2.0 ** -22):
        """The same storage model with fp32-order noise of `jitter` (default four fp32 ulps) in front of every store."""
        return Bf16Storage(self.train, self.max_s, self.max_work, self.keep_p_max_s, jitter, seed)

    def weight(self, w):
        return _RoundFwd.apply(w, None)       # (a weight copy is a plain cast: no evaluation-order noise in front of it)

    def prob(self, p):
        return _RoundFwd.apply(p, self)

    def both(self, x):
        return _RoundBoth.apply(x, self)

    def grad(self, x):
        return _RoundGrad.apply(x, self)

    def act_store(self, z):
        return _ActStore.apply(z, self)

    def p_is_rounded(self, n, heads, S, dk):
        small = S <= self.max_s and dk % 16 == 0 and n * heads * S * S <= self.max_work
        if small and self.train:
            small = S <= self.keep_p_max_s
        return self.train or not small      # the P-keeping forms store bf16 probabilities; only the no-grad small core does not


# --------------------------------------------------------------------------- #
# layers
# --------------------------------------------------------------------------- #
def silu(x):
    return x * torch.sigmoid(x)


def sinusoid(t, D):
    """custom_layers.py:84-90."""
    half = D // 2
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10_000) / (half - 1)))
    a = t[:, None] * f[None, :]
    return torch.cat((a.sin(), a.cos()), dim=1)


def mlp4(P, prefix, x):
    for i in (0, 2, 4):
        x = silu(F.linear(x, P[f"{prefix}.{i}.weight"], P[f"{prefix}.{i}.bias"]))
    return F.linear(x, P[f"{prefix}.6.weight"], P[f"{prefix}.6.bias"])


def cond_embedding(P, cfg, t, cond=None):
    emb = mlp4(P, "cond_emb.time_layer", sinusoid(t, cfg["time_dim"]))
    if cfg["cond_dim"] is not None:
        emb = emb + mlp4(P, "cond_emb.cond_layer", cond)
    return emb


def adagn(P, name, x, emb):
    """s * GN(x) + s  with s = y_scale(emb); y_shift is a dead parameter."""
    xg = F.group_norm(x, 32, P[name + ".group_norm.weight"], P[name + ".group_norm.bias"], 1e-5)
    s = F.linear(emb, P[name + ".y_scale.weight"], P[name + ".y_scale.bias"])[:, :, None, None]
    return s * xg + s


def conv_block(P, name, x, emb=None, act=True, Q=None, edge=None):
    """Q: Bf16Storage or None (fp32, the reference's arithmetic).  edge: "stem" (fp32 weights and image) / "head" (fp32
    result, its input gradient stored bf16) on bf16 rows."""
    w, b = P[name + ".conv_layer.0.weight"], P[name + ".conv_layer.0.bias"]
    if Q is None:
        x = F.conv2d(x, w, b, padding=1)
        if act:
            x = silu(x)
    elif edge == "stem":
        x = Q.act_store(_StemConv.apply(x, w, b) if Q.train else F.conv2d(x, w, b, padding=1))
    elif edge == "head":
        return Q.grad(F.conv2d(x, Q.weight(w), b, padding=1))
    else:
        x = F.conv2d(x, Q.weight(w), b, padding=1)
        x = Q.act_store(x) if act else Q.both(x)
    if emb is not None:
        x = adagn(P, name + ".adagn", x, emb)      # statistics of the stored values; the caller stores the modulated tensor
    return x


def res_block(P, name, x, emb, Q=None):
    y = conv_block(P, name + ".conv_block_1", x, emb, Q=Q)
    if Q is not None and emb is not None:
        y = Q.both(y)                              # sdm_affine_residual_bf16 materialises the affine ahead of the second conv
    y = conv_block(P, name + ".conv_block_2", y, emb, Q=Q)
    return y + x if Q is None else Q.both(y + x)


def attention(P, name, x, heads, dk, Q=None):
    N, C, H, W = x.shape
    d = dk or C
    xs = x.reshape(N, C, H * W).transpose(1, 2)                    # [N,S,C]
    wp, wo = P[name + ".projection.weight"], P[name + ".output.weight"]
    if Q is not None:
        wp, wo = Q.weight(wp), Q.weight(wo)
    qkv = F.linear(xs, wp, P[name + ".projection.bias"])
    if Q is not None:
        qkv = Q.both(qkv)
    qkv = qkv.reshape(N, H * W, heads, 3 * d)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    q, k, v = (z.permute(0, 2, 1, 3) for z in (q, k, v))           # [N,h,S,d]
    s = torch.matmul(q, k.transpose(-1, -2)) * d ** -0.5           # [N,h,i,j]
    if Q is not None:
        s = Q.grad(s)                                              # dS of the column softmax is stored bf16
    p = torch.softmax(s, dim=2)                                    # over queries i
    if Q is not None and Q.p_is_rounded(N, heads, H * W, d):
        p = Q.prob(p)                                              # bf16 probabilities; dP stays fp32
    o = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(N, H * W, heads * d)
    if Q is not None:
        o = Q.both(o)
    o = F.linear(o, wo, P[name + ".output.bias"]) + xs
    if Q is not None:
        o = Q.both(o)
    return o.transpose(1, 2).reshape(N, C, H, W)


def unet_block(P, cfg, name, x, emb, use_attn, up, Q=None):
    for r in range(cfg["num_resnet_blocks"]):
        x = res_block(P, f"{name}.res_layers.{r}", x, emb, Q=Q)
        if use_attn:
            x = attention(P, f"{name}.attn_layers.{r}", x, cfg["num_heads"], cfg["dim_per_head"], Q=Q)
    w, b = P[f"{name}.out_layer.conv_layer.0.weight"], P[f"{name}.out_layer.conv_layer.0.bias"]
    if Q is not None:
        w = Q.weight(w)
    if up:
        x = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    else:
        x = F.conv2d(x, w, b, stride=2, padding=1)
    return silu(x) if Q is None else Q.act_store(x)


def unet_forward(P, cfg, x, t=None, cond=None, storage=None):
    """models/U_Net.py:147-173.  storage: None / torch.float32 = the reference's fp32 arithmetic; torch.bfloat16 or a
    Bf16Storage = the bf16 storage model of the HIP path (inference unless Bf16Storage(train=True))."""
    cfg = full_cfg(cfg)
    Q = None if storage in (None, torch.float32) else (storage if isinstance(storage, Bf16Storage) else Bf16Storage())
    L = cfg["num_layers"]
    emb = cond_embedding(P, cfg, t, cond) if cfg["time_dim"] is not None else None
    x = conv_block(P, "in_layer.0", x, Q=Q, edge="stem")
    x = conv_block(P, "in_layer.1", x, Q=Q)
    skips = []
    for k in range(L):
        x = unet_block(P, cfg, f"down_layers.{k}", x, emb, k in cfg["attn_layers"], False, Q=Q)
        skips.append(x)
    x = conv_block(P, "middle_layer.0", x, Q=Q)
    x = conv_block(P, "middle_layer.1", x, Q=Q)
    for idx, k in enumerate(range(L - 1, -1, -1)):
        skip = skips.pop()
        x = torch.cat((x, skip if Q is None else Q.grad(skip)), dim=1)
        x = unet_block(P, cfg, f"up_layers.{idx}", x, emb, k in cfg["attn_layers"], True, Q=Q)
    x = conv_block(P, "out_layers.0", x, Q=Q)
    x = conv_block(P, "out_layers.1", x, act=False, Q=Q, edge="head")
    if cfg["image_recon"]:
        x = torch.tanh(x)
    return x


# --------------------------------------------------------------------------- #
# schedules (degraders.py)
# --------------------------------------------------------------------------- #
class LinearSchedule:
    def __init__(self, beta_1, beta_T, T):
        self.beta = torch.linspace(beta_1, beta_T, int(T + 1))
        self.alpha = 1 - self.beta
        self.abar = torch.cumprod(self.alpha, dim=0)

    def params(self, t):
        return self.beta[t], self.alpha[t], self.abar[t]

    def alpha_bar(self, t):
        return self.abar[t]


class CosineSchedule:
    def __init__(self, T=1000):
        self.T = T
        self.s = 0.008

    def alpha_bar(self, t):
        f = lambda u: torch.cos(((u / self.T) + self.s) / (1 + self.s) * (math.pi / 2)) ** 2
        return f(t) / f(torch.zeros_like(t))

    def params(self, t):
        ab = self.alpha_bar(t)
        beta = torch.clip(1 - ab / self.alpha_bar(t - 1), 0.001, 0.999)
        return beta, 1 - beta, ab


def q_sample(sched, x0, t, eps):
    ab = sched.alpha_bar(t)[:, None, None, None]
    return ab ** 0.5 * x0 + (1 - ab) ** 0.5 * eps


# --------------------------------------------------------------------------- #
# samplers with injected noise (diffusion_sampling_algorithms.py)
# --------------------------------------------------------------------------- #
def ddpm_sample(P, cfg, sched, x_t, noises, min_noise=1, max_noise=1000, labels=None):
    """noises[i] is the z drawn at loop iteration i (ignored when noise_step <= 1)."""
    traj = []
    with torch.no_grad():
        for i, step in enumerate(range(max_noise, min_noise - 1, -1)):
            t = torch.tensor([step])
            beta, alpha, abar = sched.params(t)
            eps = unet_forward(P, cfg, x_t, t, labels)
            z = noises[i] if step > 1 else 0
            x_t = (1 / alpha ** 0.5) * (x_t - ((1 - alpha) / (1 - abar) ** 0.5) * eps) + beta ** 0.5 * z
            traj.append(x_t)
    return x_t, traj


# --------------------------------------------------------------------------- #
# the DDPM step's noise as the HIP path draws it (sdm_ddpm_update): Philox4x32-10 + Box-Muller.
# The reference draws torch.randn_like(x_t) (diffusion_sampling_algorithms.py:43) on whatever device it runs on; a CPU
# and a GPU generator give different streams anyway, so the stream itself is not a parity quantity -- what is pinned
# here is the generator (Random123's published known-answer vectors, tests/test_oracle_golden.py) and, against this
# restatement, the kernel's bits -> normals mapping (tests/test_gpu_parity.py).
# --------------------------------------------------------------------------- #
def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] -> uint32 [..., 4] (Salmon et al., SC'11)."""
    import numpy as np
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k = [np.asarray(key[..., i], dtype=np.uint64) for i in range(2)]
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), \
        np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return np.stack(c, axis=-1).astype(np.uint32)


def philox_normal(seed, step, n):
    """The n standard normals sdm_ddpm_update draws for (seed, step): element i comes from Philox counter
    {i // 4 (lo, hi), step, 0}, key {seed lo, seed hi}; words (0, 1) and (2, 3) go through Box-Muller with
    u1 = ((w >> 8) + 1) / 2^24, u2 = (w >> 8) / 2^24.  float64 arithmetic (the kernel's is float32)."""
    import numpy as np
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.stack([q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.full_like(q, np.uint64(step & 0xFFFFFFFF)),
                    np.zeros_like(q)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64), (q.shape[0], 2))
    r = philox4x32_10(ctr, key).astype(np.float64)

    def bm(a, b):
        u1 = (np.floor(a / 256.0) + 1.0) / 16777216.0
        u2 = np.floor(b / 256.0) / 16777216.0
        rad = np.sqrt(-2.0 * np.log(u1))
        return rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)

    z0, z1 = bm(r[:, 0], r[:, 1])
    z2, z3 = bm(r[:, 2], r[:, 3])
    return np.stack([z0, z1, z2, z3], axis=-1).reshape(-1)[:n]


def ddim_steps(min_noise, max_noise, step):
    steps = list(range(max_noise, min_noise - 1, -step))
    if min_noise not in steps:
        steps.append(min_noise)
    return steps


def ddim_sample(P, cfg, sched, x_t, min_noise=1, max_noise=1000, step=10, labels=None, cond_img=None):
    steps = ddim_steps(min_noise, max_noise, step)
    traj = []
    with torch.no_grad():
        for c, s in enumerate(steps):
            t = torch.tensor([s])
            eps = unet_forward(P, cfg, x_t if cond_img is None else torch.cat((x_t, cond_img), dim=1), t, labels)
            abar = sched.params(t)[2]
            x0 = (1 / abar ** 0.5) * (x_t - (1 - abar) ** 0.5 * eps)
            if c < len(steps) - 1:
                ab1 = sched.params(torch.tensor([steps[c + 1]]))[2]
                x_t = ab1 ** 0.5 * x0 + (1 - ab1) ** 0.5 * eps      # eta = 0 -> sigma = 0
                traj.append(x_t)
    return (x0 if steps[-1] == 1 else x_t), traj


def cold_sample(P, cfg, sched, x_t, noise, cond_img, min_noise=1, max_noise=1000, step=10, labels=None):
    """diffusion_sampling_algorithms.py:150-217 (x0-prediction sampler of the SR path)."""
    steps = ddim_steps(min_noise, max_noise, step)
    with torch.no_grad():
        for c, s in enumerate(steps):
            t = torch.tensor([s])
            x0 = unet_forward(P, cfg, torch.cat((x_t, cond_img), dim=1) if cond_img is not None else x_t, t, labels)
            if c < len(steps) - 1:
                tm1 = torch.tensor([steps[c + 1]])
                x_t = x_t - q_sample(sched, x0, t, noise) + q_sample(sched, x0, tm1, noise)
    return x0


def area_resize(x, size):
    """F.interpolate(mode="area") (train_SR_diffusion.py:321-328)."""
    return F.interpolate(x, size=size, mode="area")


def sr_batch(sched, x0, t, eps, lr_dim, sr_dim, cond_t):
    """(network input, regression target) of the SR training step (train_SR_diffusion.py:318-374)."""
    lr = area_resize(area_resize(x0, (lr_dim, lr_dim)), (sr_dim, sr_dim))
    x_in = torch.cat((q_sample(sched, x0, t, eps), q_sample(sched, lr, torch.tensor([cond_t]), eps)), dim=1)
    return x_in, x0 - lr


# --------------------------------------------------------------------------- #
# training step (train_diffusion.py:307-366)
# --------------------------------------------------------------------------- #
def loss_and_grads(P, cfg, sched, x0, t, eps, labels=None):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    x_t = q_sample(sched, x0, t, eps)
    out = unet_forward(leaves, cfg, x_t, t, labels)
    loss = F.mse_loss(out, eps)
    loss.backward()
    return loss.detach(), out.detach(), {k: v.grad for k, v in leaves.items()}


class OracleTrainer:
    """Adam(lr, betas=(0.5, 0.999)) over the live parameters, stock torch CPU."""

    def __init__(self, P, cfg, sched, lr=2e-5):
        self.cfg, self.sched = cfg, sched
        self.P = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr, betas=(0.5, 0.999))

    def step(self, x0, t, eps, labels=None, target="eps", cond_img=None):
        """target "eps": train_diffusion.py:343-352; "x0": cold diffusion, train_noise_cold_diffusion.py:335-342;
        cond_img: doodle, the label image concatenated to x_t (train_doodle_diffusion.py:304)."""
        self.opt.zero_grad()
        x_t = q_sample(self.sched, x0, t, eps)
        if cond_img is not None:
            x_t = torch.cat((x_t, cond_img), dim=1)
        out = unet_forward(self.P, self.cfg, x_t, t, labels)
        loss = F.mse_loss(out, eps if target == "eps" else x0)
        loss.backward()
        self.opt.step()
        return float(loss.detach())
