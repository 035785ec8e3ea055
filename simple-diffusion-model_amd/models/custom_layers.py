"""U-Net building blocks executing on libsdm_hip (MI355X).

Drop-in for the reference's ``models/custom_layers.py``: same class names, constructor
arguments and ``state_dict`` keys, but the modules only *own parameters*;
all arithmetic is launched through ``sdm_hip.ops`` (C ABI, hand-written HIP).

Layout decisions that differ from the reference (and why):
  * activations travel as NHWC pixel rows (``ops.Act``) so a conv tap is a contiguous-K row
    gather and attention's [N,S,C] view is free;
  * conv weights live tap-major ``[ky][kx][c_out][c_in]`` in memory; the Parameter is a strided
    *view* with the PyTorch shape, so checkpoints, ``load_state_dict`` and Adam see the usual
    tensors while the kernels read (and write gradients) with unit stride;
  * GroupNorm+AdaGN is folded into a per-(image,channel) affine applied by the *consumer* conv
    while it loads its input tile (padding stays exactly zero).

Behavioural quirks reproduced on purpose (reference file:line):
  AdaGN shift == scale (custom_layers.py:41); attention ``norm`` is a dead parameter (:113);
  softmax over the query axis (:147); AdaGN groups fixed at 32 (:313-316,:220).

Every block has ``run(ctx, act, ...)`` (NHWC rows in/out, used by U_Net) and, when the context is
in training mode, leaves a record in ``ctx.rec[self]`` that its ``backward`` consumes.  ``forward``
(NCHW in/out) exists for standalone use of a block and for block-level tests.
"""
import math
import os
from enum import Enum

import torch
import torch.nn as nn

from sdm_hip import ops
from sdm_hip.ops import ACT_NONE, ACT_SILU, Act


class UNetBlockType(Enum):
    UP = 0
    DOWN = 1


class RunCtx:
    """Per-forward execution context.  ``train`` keeps what backward needs in ``rec`` (module -> record)."""

    def __init__(self, train=False, bf16=False):
        self.train = train
        self.bf16 = bf16                   # bf16 storage: sampling forwards, and the bf16 training path (fp32 master
                                           # weights / weight gradients / optimizer, everything summed in fp32)
        self.rec = {}
        self.emb = None
        self.demb = None
        self.mods = None      # {AdaGN: modulation rows} when the net evaluated all y_scale projections in one launch
        self.dmods = None     # {AdaGN: gradient rows} consumed by the batched y_scale data gradient
        self.bank = None      # the y_scale job tables this forward used (ops.LinearBank.tables())


MATERIALIZE_TRAIN = [os.environ.get("SDM_TRAIN_MATERIALIZE", "1") != "0"]
FUSE_GN = [os.environ.get("SDM_FUSE_GN", "1") != "0"]       # fused conv-block tail in no-grad forwards
FUSE_GN_TRAIN = [os.environ.get("SDM_FUSE_GN_TRAIN", "1") != "0"]   # ... and in fp32 training steps on small layers
FUSE_GN_BF16 = [os.environ.get("SDM_FUSE_GN_BF16", "1") != "0"]   # bf16 rows (small outputs, as fp32): N=1 step 0.826 -> 0.776 ms
FUSE_GN_MAX_BYTES = 8 << 20                                   # ... for conv outputs up to 8 MB (see ResidualBlock._fuse_this)


def bf16_weight(mod, key, src_fn, transposed=False, pad_to=1):
    """bf16 copy of a contraction weight, refreshed IN PLACE when the fp32 master changed (in-place torch updates bump
    ``_version``; the fused optimizer, which writes through raw pointers, bumps ops.WEIGHT_EPOCH) -- the copy keeps its
    address, so captured graphs stay valid.  ``transposed``: [T, R, C] -> [T, C, R] (what a bf16 data gradient reads)."""
    w = mod._parameters["weight"]
    tag = (w.data_ptr(), w._version, ops.WEIGHT_EPOCH[0])
    cache = mod.__dict__.setdefault("_bf16_cache", {})
    hit = cache.get(key)
    if hit is not None and len(hit) == 3:
        # a copy the fused optimizer writes itself, next to the fp32 master (FusedAdam.adopt_bf16_copies): in step with the
        # weight for as long as that optimizer is the only raw writer and torch has not touched the parameter
        token, switches = hit[2]
        if hit[0][:2] == tag[:2] and token is ops._LAST_WRITER[0] and switches == ops.WRITER_SWITCHES[0]:
            return hit[1]
    if hit is None or hit[0] != tag:
        keep = hit[1] if (hit is not None and hit[0][0] == tag[0]) else None
        if transposed:
            cache[key] = hit = (tag, ops.to_bf16_t(src_fn(), out=keep, pad_to=pad_to))
        else:
            cache[key] = hit = (tag, ops.to_bf16(src_fn(), out=keep))
    return hit[1]


def to_rows(x_nchw):
    n, c, h, w = x_nchw.shape
    return Act(x_nchw.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous(), n, h, w)


def to_nchw(a):
    return a.t.reshape(a.n, a.h, a.w, a.c).permute(0, 3, 1, 2).contiguous()


def param_grad_like(p):
    """ZERO-FILLED gradient buffer with the parameter's own memory layout (tap-major for conv weights).
    Backward kernels accumulate into gradient buffers (split-K atomics, column sums), so they start from zero."""
    return torch.zeros(p.numel(), dtype=p.dtype, device=p.device).as_strided(p.shape, p.stride())


class Swish(nn.Module):
    """x * sigmoid(x).  Inside the U-Net it is fused into the producing kernel's epilogue."""

    def forward(self, x):
        return ops.silu(x)


class TapLinear(nn.Module):
    """nn.Linear parameters (weight [out, in], bias [out]) drawn by torch's default initialiser."""

    def __init__(self, in_features, out_features):
        super().__init__()
        ref = nn.Linear(in_features, out_features)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())
        self.in_features, self.out_features = in_features, out_features

    def weight_for(self, ctx):
        return bf16_weight(self, "w", lambda: self.weight) if ctx.bf16 else self.weight

    def weight_t_for(self, ctx):
        """What the data gradient contracts against: the fp32 weight itself (read K-major), or -- bf16 -- the transposed
        copy [in, out]."""
        if not ctx.bf16:
            return self.weight
        return bf16_weight(self, "wt", lambda: self.weight.detach()[None], transposed=True)[0]


class TapNorm(nn.Module):
    """nn.GroupNorm affine parameters."""

    def __init__(self, groups, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.groups = groups


class TapConv(nn.Module):
    """Conv2d(k3) / ConvTranspose2d(k4,s2,p1) parameters stored tap-major.  Initial values are drawn by the
    stock torch module so that a seeded construction matches the reference draw for draw."""

    def __init__(self, cin, cout, kind):
        super().__init__()
        self.kind, self.cin, self.cout = kind, cin, cout
        if kind == "ct4":
            ref = nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1)
            base = ref.weight.detach().permute(2, 3, 1, 0).contiguous()        # [4,4,cout,cin]
            self.weight = nn.Parameter(base.permute(3, 2, 0, 1))               # shape (cin,cout,4,4)
        else:
            ref = nn.Conv2d(cin, cout, kernel_size=3, stride=1 if kind == "c3s1" else 2, padding=1)
            base = ref.weight.detach().permute(2, 3, 0, 1).contiguous()        # [3,3,cout,cin]
            self.weight = nn.Parameter(base.permute(2, 3, 0, 1))               # shape (cout,cin,3,3)
        self.bias = nn.Parameter(ref.bias.detach().clone())

    def _tapview(self, w):
        return w.permute(2, 3, 1, 0) if self.kind == "ct4" else w.permute(2, 3, 0, 1)

    def taps(self):
        """[taps, c_out, c_in] contiguous view of the weight storage (re-laid out once if something replaced it)."""
        w = self._parameters["weight"]
        cached = self.__dict__.get("_taps_cache")
        if cached is not None and cached[0] == w.data_ptr() and cached[1] == w.device:
            return cached[2]
        v = self._tapview(w.data)
        if not v.is_contiguous():
            base = v.contiguous()
            self.weight.data = base.permute(3, 2, 0, 1) if self.kind == "ct4" else base.permute(2, 3, 0, 1)
            v = base
        k = v.shape[0]
        v = v.view(k * k, self.cout, self.cin)
        self.__dict__["_taps_cache"] = (self.weight.data_ptr(), self.weight.device, v)
        return v

    def taps_for(self, ctx):
        return bf16_weight(self, "taps", self.taps) if ctx.bf16 else self.taps()

    def taps_t_for(self, ctx):
        """Weights for the data gradient: fp32 taps (read K-major) or the transposed bf16 copy [taps, c_in, c_out] (c_out
        zero-padded to a multiple of 8, the reduction granule: only the 3-channel head conv needs the padding)."""
        return bf16_weight(self, "taps_t", self.taps, transposed=True, pad_to=8) if ctx.bf16 else self.taps()

    def grad_taps(self, g):
        v = self._tapview(g)
        if not v.is_contiguous():
            raise RuntimeError("gradient buffer must share the tap-major layout of its parameter")
        k = v.shape[0]
        return v.view(k * k, self.cout, self.cin)

    def backward(self, x, y_hw, dz, grads, res=None, res2=None, need_dx=True, ctx=None):
        """dz: gradient w.r.t. this conv's pre-activation [n*ho*wo, cout] (any pitch).  Writes weight/bias
        gradients; returns the gradient w.r.t. the input rows -- already multiplied by silu'(x.z) when the
        input is itself an activation output (then it is that producer's dz).  ``ctx`` selects the bf16 path."""
        self.taps()
        wt_d = self.taps_t_for(ctx) if ctx is not None else self.taps()
        if ops.SIDE.enabled:            # weight gradient on the side stream, concurrent with the data gradient below
            with ops.SIDE.fork(dz, x.t, x.t2, x.scale, x.shift):
                ops.conv_wgrad(dz, y_hw, x, self.kind, self.grad_taps(grads[self.weight]), dbias=grads[self.bias])
        else:
            ops.conv_wgrad(dz, y_hw, x, self.kind, self.grad_taps(grads[self.weight]), dbias=grads[self.bias])
        if not need_dx:
            return None
        ncols = x.c1 if x.t2 is not None else 0
        return ops.conv_dgrad(dz, y_hw, wt_d, self.kind, (x.h, x.w), x.n, res=res, res2=res2,
                              dact_z=x.z, dact_ncols=ncols)


class AdaGN(nn.Module):
    def __init__(self, emb_dim, out_dim, groups=32):
        super().__init__()
        self.y_scale = TapLinear(emb_dim, out_dim)
        self.y_shift = TapLinear(emb_dim, out_dim)       # dead parameter, kept for checkpoint compatibility
        self.group_norm = TapNorm(groups, out_dim)

    def modulation(self, emb):
        return ops.linear_small(emb, self.y_scale.weight, self.y_scale.bias)

    def fold(self, y, mod, want_stats=True):
        """-> (scale, shift, mean, rstd) with scale*y+shift == s*GN(y)+s."""
        return ops.gn_affine(y, self.group_norm.weight, self.group_norm.bias, mod, self.group_norm.groups,
                             want_stats=want_stats)

    def backward(self, ctx, dout, y, mod, mean, rstd, grads):
        """dout: gradient w.r.t. the modulated tensor.  Returns dz of the producing conv."""
        gn = self.group_norm
        gw, gb = grads[gn.weight], grads[gn.bias]
        batched = ctx.dmods is not None
        dz, dmod = ops.gn_adagn_bwd(dout, y.t, y.z, y.n, y.h * y.w, gn.weight, gn.bias, mod, mean, rstd, gw, gb, gn.groups,
                                    dmod=ctx.dmods[self] if batched else None)
        dW, db = grads[self.y_scale.weight], grads[self.y_scale.bias]
        # weight/bias gradients now (the block's gradients must be complete when the block ends: DP buckets);
        # the embedding gradient of ALL projections is one batched launch at the end of the backward
        if batched and ops.SIDE.enabled:          # weight / bias gradient only: side stream
            with ops.SIDE.fork(dmod, ctx.emb):
                ops.linear_small_bwd(dmod, ctx.emb, self.y_scale.weight, dx=None, dW=dW, db=db)
        else:
            ops.linear_small_bwd(dmod, ctx.emb, self.y_scale.weight, dx=None if batched else ctx.demb, dx_accumulate=True,
                                 dW=dW, db=db)
        return dz

    def forward(self, x, emb):
        y = to_rows(x)
        scale, shift, _, _ = self.fold(y, self.modulation(emb.contiguous()))
        zero = Act(torch.zeros_like(y.t), y.n, y.h, y.w)
        return to_nchw(ops.affine_residual(y, scale, shift, zero))


class ConditionalEmbedding(nn.Module):
    def __init__(self, time_dim, cond_dim=None):
        super().__init__()
        self.time_dim, self.cond_dim = time_dim, cond_dim

        def mlp(first_in):       # Linear at 0,2,4,6 (the reference interleaves Swish modules at 1,3,5)
            return nn.ModuleList([TapLinear(first_in, time_dim), nn.Identity(), TapLinear(time_dim, time_dim),
                                  nn.Identity(), TapLinear(time_dim, time_dim), nn.Identity(),
                                  TapLinear(time_dim, time_dim)])

        self.time_layer = mlp(time_dim)
        self.cond_layer = mlp(cond_dim) if cond_dim is not None else None
        half = time_dim // 2
        # frequency table exactly as the reference computes it (custom_layers.py:84-88)
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10_000) / (half - 1)))
        self.register_buffer("_freq", freq, persistent=False)

    def run(self, ctx, t, cond=None):
        """time_layer(sinusoid(t)) + cond_layer(cond) with the reference's broadcasting (custom_layers.py:92-98): t of
        shape [1] against labels [n, cond_dim] (the samplers at every checkpoint plot) or t [N] against one label row."""
        rec = {} if ctx.train else None
        t = t.reshape(-1).to(torch.int64).contiguous()
        c = None
        if self.cond_layer is not None:
            c = cond.to(torch.float32)
            if c.dim() == 1:
                c = c[None, :]
            T, n = t.shape[0], c.shape[0]
            if T != n and T != 1 and n != 1:
                raise RuntimeError(f"The size of tensor a ({T}) must match the size of tensor b ({n}) at non-singleton "
                                   "dimension 0")
            if n == 1 and T > 1:
                c = c.expand(T, -1)                  # one label row for the whole batch
            if rec is not None and T == 1 and n > 1:
                t = t.expand(n).contiguous()         # training keeps one embedding row per image on both branches
        pe = ops.sinusoid(t, self._freq, self.time_dim)
        emb = self._mlp(self.time_layer, pe, None, rec, "time")
        if c is not None:
            emb = self._mlp(self.cond_layer, c.contiguous(), emb, rec, "cond")     # emb may be ONE row: broadcast add
        if rec is not None:
            ctx.rec[self] = rec
        return emb

    def time_rows(self, t):
        """time_layer(sinusoid(t)) for a vector of timesteps -> [len(t), D] (one row per sampling step: the samplers
        tabulate the embedding of a whole run once, custom_layers.py:82-95 depends on t only)."""
        t = t.reshape(-1).to(torch.int64).contiguous()
        return self._mlp(self.time_layer, ops.sinusoid(t, self._freq, self.time_dim), None, None, "time")

    def cond_rows(self, cond):
        """cond_layer(cond) -> [n, D] (custom_layers.py:96-98 without the time addend)."""
        c = cond.to(torch.float32)
        if c.dim() == 1:
            c = c[None, :]
        return self._mlp(self.cond_layer, c.contiguous(), None, None, "cond")

    @staticmethod
    def _mlp(layers, x, add, rec, tag):
        xs, zs = [x], []
        for i in (0, 2, 4):
            if rec is not None:
                x, z = ops.linear_small(x, layers[i].weight, layers[i].bias, ACT_SILU, save_z=True)
                zs.append(z)
            else:
                x = ops.linear_small(x, layers[i].weight, layers[i].bias, ACT_SILU)
            xs.append(x)
        out = ops.linear_small(x, layers[6].weight, layers[6].bias, ACT_NONE, add=add)
        if rec is not None:
            rec[tag] = (xs, zs)
        return out

    def backward(self, ctx, demb, grads):
        """demb: [N, D] gradient of the summed embedding."""
        for tag, layers in (("time", self.time_layer), ("cond", self.cond_layer)):
            if layers is None:
                continue
            xs, zs = ctx.rec[self][tag]
            g = demb
            for j, i in enumerate((6, 4, 2, 0)):
                lin = layers[i]
                gw, gb = grads[lin.weight], grads[lin.bias]
                x_in = xs[3 - j]
                if i == 0:
                    ops.linear_small_bwd(g, x_in, lin.weight, dx=None, dW=gw, db=gb)
                else:
                    dx = torch.zeros_like(x_in)
                    ops.linear_small_bwd(g, x_in, lin.weight, dx=dx, dW=gw, db=gb)
                    g = ops.silu_bwd(dx, zs[2 - j])

    def forward(self, t, cond=None):
        return self.run(RunCtx(False), t, cond)


class AttentionBlock(nn.Module):
    def __init__(self, channels, heads=1, d_k=None, groups=32):
        super().__init__()
        if d_k is None:
            d_k = channels
        self.norm = TapNorm(groups, channels)            # constructed, never applied (reference :113 vs :127-163)
        self.projection = TapLinear(channels, heads * d_k * 3)
        self.output = TapLinear(heads * d_k, channels)
        self.scale = d_k ** -0.5
        self.heads, self.d_k = heads, d_k

    def run(self, ctx, x):
        S = x.h * x.w
        qkv = ops.linear_rows(x.t, self.projection.weight_for(ctx), self.projection.bias)
        if ctx.bf16:
            if ctx.train:
                res, P = ops.attention_core_fwd_bf16(qkv, x.n, S, self.heads, self.d_k, keep_p=True)
                ctx.rec[self] = (x, qkv, P, res)
            else:
                res = ops.attention_core_fwd_bf16(qkv, x.n, S, self.heads, self.d_k)
            out = ops.linear_rows(res, self.output.weight_for(ctx), self.output.bias, res=x.t)
            return Act(out, x.n, x.h, x.w)
        res, P = ops.attention_core_fwd(qkv, x.n, S, self.heads, self.d_k, keep_p=ctx.train)
        out = ops.linear_rows(res, self.output.weight, self.output.bias, res=x.t)
        if ctx.train:
            ctx.rec[self] = (x, qkv, P, res)
        return Act(out, x.n, x.h, x.w)

    def backward(self, ctx, dO, grads):
        """dO: gradient w.r.t. the block output rows; returns the gradient w.r.t. the input rows."""
        x, qkv, P, res = ctx.rec[self]
        S = x.h * x.w
        if ops.SIDE.enabled:
            with ops.SIDE.fork(dO, res):
                ops.linear_rows_wgrad(dO, res, grads[self.output.weight], dbias=grads[self.output.bias])
        else:
            ops.linear_rows_wgrad(dO, res, grads[self.output.weight], dbias=grads[self.output.bias])
        dres = ops.linear_rows_dgrad(dO, self.output.weight_t_for(ctx))
        core_bwd = ops.attention_core_bwd_bf16 if ctx.bf16 else ops.attention_core_bwd
        dqkv = core_bwd(dres, qkv, P, x.n, S, self.heads, self.d_k)
        if ops.SIDE.enabled:
            with ops.SIDE.fork(dqkv, x.t):
                ops.linear_rows_wgrad(dqkv, x.t, grads[self.projection.weight], dbias=grads[self.projection.bias])
        else:
            ops.linear_rows_wgrad(dqkv, x.t, grads[self.projection.weight], dbias=grads[self.projection.bias])
        return ops.linear_rows_dgrad(dqkv, self.projection.weight_t_for(ctx), res=dO)

    def forward(self, x, t=None):
        return to_nchw(self.run(RunCtx(False), to_rows(x)))


class _ConvHolder(nn.Module):
    """``conv_layer.0`` naming shim: the reference wraps its conv in nn.Sequential(conv, Swish())."""

    def __init__(self, conv):
        super().__init__()
        self.add_module("0", conv)

    @property
    def conv(self):
        return self._modules["0"]


class _ResampleBlock(nn.Module):
    kind = None

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv_layer = _ConvHolder(TapConv(in_channels, out_channels, self.kind))

    def run(self, ctx, x, emb=None):
        c = self.conv_layer.conv
        y = ops.conv_fwd(x, c.taps_for(ctx), c.bias, self.kind, ACT_SILU, save_z=ctx.train)
        if ctx.train:
            ctx.rec[self] = (x, y)
        return y

    def backward(self, ctx, dz, grads, res=None, res2=None):
        x, y = ctx.rec[self]
        return self.conv_layer.conv.backward(x, (y.h, y.w), dz, grads, res=res, res2=res2, ctx=ctx)

    def forward(self, x, emb=None):
        return to_nchw(self.run(RunCtx(False), to_rows(x)))


class UpsampleBlock(_ResampleBlock):
    kind = "ct4"        # ConvTranspose2d(k4, s2, p1) + Swish


class DownsampleBlock(_ResampleBlock):
    kind = "c3s2"       # Conv2d(k3, s2, p1) + Swish


class UNet_ConvBlock(nn.Module):
    """conv3x3 -> [SiLU] -> [AdaGN].  ``run`` returns the raw conv output plus the folded affine (if any);
    the consumer applies the affine while loading."""

    def __init__(self, in_channels, out_channels, use_activation=True, emb_dim=None, groups=32):
        super().__init__()
        self.conv_layer = _ConvHolder(TapConv(in_channels, out_channels, "c3s1"))
        self.use_activation = use_activation
        if emb_dim is not None:
            self.adagn = AdaGN(emb_dim, out_channels, groups=groups)

    def run(self, ctx, x, emb=None):
        c = self.conv_layer.conv
        act = ACT_SILU if self.use_activation else ACT_NONE
        has_gn = emb is not None and hasattr(self, "adagn")
        y = ops.conv_fwd(x, c.taps_for(ctx), c.bias, "c3s1", act, save_z=ctx.train and self.use_activation,
                         gn_stats=has_gn and ctx.bf16)
        if emb is None or not hasattr(self, "adagn"):
            if ctx.train:
                ctx.rec[self] = (x, y, None, None, None)
            return y, None
        mod = ctx.mods[self.adagn] if ctx.mods is not None else self.adagn.modulation(emb)
        scale, shift, mean, rstd = self.adagn.fold(y, mod, want_stats=ctx.train)
        if ctx.train:
            ctx.rec[self] = (x, y, mod, mean, rstd)
        return y, (scale, shift)

    def run_fused(self, ctx, x, emb, residual=None):
        """conv -> SiLU -> AdaGN (+ residual) with the tail in one launch (sdm_gn_apply_fused); no-grad fp32 forwards."""
        c = self.conv_layer.conv
        act = ACT_SILU if self.use_activation else ACT_NONE
        mod = ctx.mods[self.adagn] if ctx.mods is not None else self.adagn.modulation(emb)
        y = ops.conv_fwd(x, c.taps_for(ctx), c.bias, "c3s1", act, defer=True)   # nothing may touch the scratch until consumed
        gn = self.adagn.group_norm
        return ops.gn_apply_fused(y, gn.weight, gn.bias, mod, gn.groups, residual=residual)

    def run_fused_train(self, ctx, x, emb, residual=None):
        """run_fused inside a training step (fp32 rows, small layers): same single-launch tail, which also leaves the
        conv's rows, its pre-activation and the GroupNorm statistics for the backward."""
        c = self.conv_layer.conv
        act = ACT_SILU if self.use_activation else ACT_NONE
        mod = ctx.mods[self.adagn] if ctx.mods is not None else self.adagn.modulation(emb)
        y = ops.conv_fwd(x, c.taps_for(ctx), c.bias, "c3s1", act, save_z=self.use_activation, defer=True)
        gn = self.adagn.group_norm
        out, ya, mean, rstd = ops.gn_apply_fused(y, gn.weight, gn.bias, mod, gn.groups, residual=residual, train=True)
        ctx.rec[self] = (x, ya, mod, mean, rstd)
        return out

    def backward(self, ctx, g, grads, res=None, res2=None):
        """g: gradient w.r.t. this block's output -- the modulated tensor when AdaGN is present, else the conv's
        pre-activation gradient (dz)."""
        x, y, mod, mean, rstd = ctx.rec[self]
        dz = g if mod is None else self.adagn.backward(ctx, g, y, mod, mean, rstd, grads)
        return self.conv_layer.conv.backward(x, (y.h, y.w), dz, grads, res=res, res2=res2, ctx=ctx)

    def forward(self, x, emb=None):
        y, aff = self.run(RunCtx(False), to_rows(x), None if emb is None else emb.contiguous())
        if aff is None:
            return to_nchw(y)
        zero = Act(torch.zeros_like(y.t), y.n, y.h, y.w)
        return to_nchw(ops.affine_residual(y, aff[0], aff[1], zero))


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, use_activation=True, emb_dim=None, groups=32):
        super().__init__()
        self.conv_block_1 = UNet_ConvBlock(in_channels, out_channels, use_activation, emb_dim, groups)
        self.conv_block_2 = UNet_ConvBlock(in_channels, out_channels, use_activation, emb_dim, groups)
        if in_channels != out_channels:
            raise NotImplementedError("1x1 shortcut is never instantiated by any U_Net (custom_layers.py:307-316)")
        self.shortcut = nn.Identity()

    def _fusable(self, ctx, emb):
        """No-grad forward (fp32 or bf16 rows) with AdaGN on both conv blocks and GroupNorm groups of a multiple of 4
        channels: the tail of each conv block (split-K finishing pass, GroupNorm fold, affine, residual add) runs as ONE
        launch."""
        if ctx.train or emb is None or not FUSE_GN[0] or (ctx.bf16 and not FUSE_GN_BF16[0]):
            return False
        return True

    def _fuse_this(self, ctx, x, emb):
        """The fused tail runs ONE workgroup per (image, group) over that slice twice: it wins where the step is a chain of
        small dependent launches (N=1..4 sampling: -24 launches per step), and loses to the separate, wider kernels once
        a conv output no longer sits in L2 (measured: 128x128 batch-128 bf16 forward 35.4 -> 40.3 ms) -> small outputs only."""
        if not self._fusable(ctx, emb):
            return False
        a1, a2 = getattr(self.conv_block_1, "adagn", None), getattr(self.conv_block_2, "adagn", None)
        if a1 is None or a2 is None:
            return False
        c = self.conv_block_1.conv_layer.conv.cout
        if x.rows * c * (2 if ctx.bf16 else 4) > FUSE_GN_MAX_BYTES:
            return False
        return c % (4 * a1.group_norm.groups) == 0 and c % (4 * a2.group_norm.groups) == 0

    def _fuse_train(self, ctx, x, emb):
        """fp32 training step on small layers (the batch-4 headline: every launch of its dependent chain costs ~4.5 us):
        the materialising path below is conv [+ split-K finishing pass] + statistics + apply per conv block, the fused
        tail one launch after the conv."""
        if not ctx.train or ctx.bf16 or emb is None or not FUSE_GN_TRAIN[0] or not MATERIALIZE_TRAIN[0] or x.scale is not None:
            return False
        a1, a2 = getattr(self.conv_block_1, "adagn", None), getattr(self.conv_block_2, "adagn", None)
        if a1 is None or a2 is None:
            return False
        c = self.conv_block_1.conv_layer.conv.cout
        if x.rows * c * 4 > FUSE_GN_MAX_BYTES:
            return False
        return c % (4 * a1.group_norm.groups) == 0 and c % (4 * a2.group_norm.groups) == 0

    def run(self, ctx, x, emb=None):
        if self._fuse_this(ctx, x, emb):
            x2 = self.conv_block_1.run_fused(ctx, x, emb)
            return self.conv_block_2.run_fused(ctx, x2, emb, residual=x)
        if self._fuse_train(ctx, x, emb):
            x2 = self.conv_block_1.run_fused_train(ctx, x, emb)
            return self.conv_block_2.run_fused_train(ctx, x2, emb, residual=x)
        y1, aff1 = self.conv_block_1.run(ctx, x, emb)
        if (ctx.bf16 or (ctx.train and MATERIALIZE_TRAIN[0])) and aff1 is not None:
            # the affine is materialised: the LDS-DMA tiles cannot apply it on the fly.  bf16 rows always; in training
            # one extra pass buys the fast (LDS-DMA) kernels for this conv's forward AND its weight gradient, which
            # would otherwise both run the register-staged affine-on-load variants
            x2 = ops.affine_residual(y1, aff1[0], aff1[1], None)
        else:
            x2 = y1 if aff1 is None else Act(y1.t, y1.n, y1.h, y1.w, scale=aff1[0], shift=aff1[1])
        y2, aff2 = self.conv_block_2.run(ctx, x2, emb)
        s2, h2 = (None, None) if aff2 is None else aff2
        return ops.affine_residual(y2, s2, h2, x)

    def backward(self, ctx, dres, grads, extra=None):
        """dres: gradient w.r.t. the block output (= gradient of the modulated y2 and of the skip x).
        ``extra``: one more gradient contribution to x coming from elsewhere (U-Net skip connection)."""
        if not hasattr(self.conv_block_2, "adagn"):
            # no embedding: y2 = silu(z2) feeds the add directly
            _, y2, _, _, _ = ctx.rec[self.conv_block_2]
            dres2 = ops.silu_bwd(dres.contiguous(), y2.z)
        else:
            dres2 = dres
        dx1 = self.conv_block_2.backward(ctx, dres2, grads)      # gradient w.r.t. modulated y1 (or dz1 when no AdaGN)
        return self.conv_block_1.backward(ctx, dx1, grads, res=dres, res2=extra)

    def forward(self, x, emb=None):
        return to_nchw(self.run(RunCtx(False), to_rows(x), None if emb is None else emb.contiguous()))


class UNetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, emb_dim, num_resnet_blocks=1, use_attn=True, num_heads=1,
                 dim_per_head=None, groups=32, block_type=UNetBlockType.DOWN):
        super().__init__()
        hidden = in_channels
        self.res_layers = nn.ModuleList()
        self.attn_layers = nn.ModuleList()
        for _ in range(num_resnet_blocks):
            # `groups` is not forwarded to the residual blocks (reference :313-316) -> AdaGN always uses 32
            self.res_layers.append(ResidualBlock(hidden, hidden, emb_dim=emb_dim))
            self.attn_layers.append(AttentionBlock(hidden, heads=num_heads, d_k=dim_per_head, groups=groups)
                                    if use_attn else nn.Identity())
        self.out_layer = (DownsampleBlock(hidden, out_channels) if block_type == UNetBlockType.DOWN
                          else UpsampleBlock(hidden, out_channels))

    def run(self, ctx, x, emb=None):
        for res, attn in zip(self.res_layers, self.attn_layers):
            x = res.run(ctx, x, emb)
            if isinstance(attn, AttentionBlock):
                x = attn.run(ctx, x)
        return self.out_layer.run(ctx, x, emb)

    def backward(self, ctx, dz, grads, extra=None):
        """dz: gradient w.r.t. the out_layer pre-activation.  Returns the gradient w.r.t. the block input
        (times silu' of the input's producer where the input is an activation output)."""
        g = self.out_layer.backward(ctx, dz, grads)
        n = len(self.res_layers)
        for r in range(n - 1, -1, -1):
            attn = self.attn_layers[r]
            if isinstance(attn, AttentionBlock):
                g = attn.backward(ctx, g, grads)
            g = self.res_layers[r].backward(ctx, g, grads, extra=extra if r == 0 else None)
        return g

    def forward(self, x, emb=None):
        return to_nchw(self.run(RunCtx(False), to_rows(x), None if emb is None else emb.contiguous()))
