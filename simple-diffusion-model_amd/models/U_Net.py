"""U-Net denoiser on MI355X — drop-in for the reference's ``models.U_Net.U_Net``.

Same constructor keywords, ``forward(x, t=None, cond=None)``, ``custom_load_state_dict`` and
``state_dict`` key names/shapes as /root/reference/models/U_Net.py:11-173, so reference
checkpoints load with ``strict=True`` and the reference trainers/samplers can call it unchanged.
What differs is underneath: the forward (and, in training, the backward) is a sequence of
libsdm_hip launches over NHWC activations; see models/custom_layers.py and DESIGN.md.

Gradients: under ``torch.enable_grad()`` the whole network is ONE autograd node
(``_UNetFunction``); its backward walks the blocks in reverse with hand-written data/weight
gradient kernels and hands autograd one gradient tensor per live parameter.  The 40 dead
parameters of the reference (``adagn.y_shift.*``, ``attn_layers.*.norm.*``) receive ``None``
exactly as they do there.
"""
import os

import torch
import torch.nn as nn

from sdm_hip import ops
from sdm_hip.ops import ACT_NONE, ACT_SILU, ACT_TANH, Act

from .custom_layers import *  # noqa: F401,F403  (same star-export surface as the reference)
from .custom_layers import (AdaGN, ConditionalEmbedding, RunCtx, UNet_ConvBlock, UNetBlock, UNetBlockType,
                            param_grad_like)


HEAD_DGRAD_EDGE = [os.environ.get("SDM_HEAD_DGRAD_EDGE", "1") != "0"]   # bf16 training: head data gradient by the edge kernel

class U_Net(nn.Module):
    def __init__(self, num_resnet_blocks=5, in_channel=3, out_channel=3, time_dim=64, cond_dim=None, num_layers=5,
                 attn_layers=[2, 3, 4], num_heads=1, dim_per_head=None, groups=32, min_channel=128, max_channel=512,
                 image_recon=False):
        super().__init__()
        # argument checks and their exception types follow models/U_Net.py:29-38
        if not isinstance(num_layers, int) or not isinstance(attn_layers, list):
            raise TypeError("Invalid type!")
        if num_layers < 1:
            raise ValueError("Invalid num layer value!")
        for a in attn_layers:
            if not isinstance(a, int):
                raise ValueError("Invalid type in attention layer!")
            if a < 0 or a >= num_layers:
                raise ValueError("Invalid Attention Layer values!")

        # the reference doubles an un-clipped running value (models/U_Net.py:41-46)
        ch, c = [min_channel], min_channel
        for _ in range(num_layers):
            c *= 2
            ch.append(max_channel if c > max_channel else c)
        self.channel_layers = ch
        self.num_layers, self.image_recon = num_layers, image_recon
        # storage type of no-grad forwards; SDM_INFER_DTYPE=bf16 selects the bf16 sampling path (set_inference_dtype)
        self.inference_dtype = torch.bfloat16 if os.environ.get("SDM_INFER_DTYPE", "") == "bf16" else torch.float32
        # storage type of training forwards/backwards; SDM_TRAIN_DTYPE=bf16 selects the bf16 training path (set_training_dtype)
        self.training_dtype = torch.bfloat16 if os.environ.get("SDM_TRAIN_DTYPE", "") == "bf16" else torch.float32
        self.in_channel, self.out_channel = in_channel, out_channel

        self.cond_emb = ConditionalEmbedding(time_dim, cond_dim) if time_dim is not None else None
        self.in_layer = nn.Sequential(UNet_ConvBlock(in_channel, ch[0]), UNet_ConvBlock(ch[0], ch[0]))
        self.down_layers = nn.ModuleList(
            UNetBlock(ch[k], ch[k + 1], time_dim, num_resnet_blocks, k in attn_layers, num_heads, dim_per_head, groups,
                      UNetBlockType.DOWN) for k in range(num_layers))
        self.middle_layer = nn.Sequential(UNet_ConvBlock(ch[-1], ch[-1]), UNet_ConvBlock(ch[-1], ch[-1]))
        self.up_layers = nn.ModuleList(
            UNetBlock(ch[k + 1] * 2, ch[k], time_dim, num_resnet_blocks, k in attn_layers, num_heads, dim_per_head,
                      groups, UNetBlockType.UP) for k in range(num_layers - 1, -1, -1))
        outs = [UNet_ConvBlock(ch[0], ch[0]), UNet_ConvBlock(ch[0], out_channel, use_activation=False)]
        if image_recon:
            outs.append(nn.Tanh())
        self.out_layers = nn.Sequential(*outs)
        self._adagns = [m for m in self.modules() if isinstance(m, AdaGN)]
        self._bank = ops.LinearBank()

    # ------------------------------------------------------------------ checkpoints
    def custom_load_state_dict(self, state_dict):
        """Lenient loader (models/U_Net.py:132-145): unknown names and shape mismatches are skipped."""
        own = self.state_dict()
        for name, param in state_dict.items():
            if name not in own:
                print(f"No Layer found: {name}, skipping")
                continue
            if own[name].shape != param.data.shape:
                print(f"Skipped: {name}")
                continue
            if isinstance(param, nn.Parameter):
                param = param.data
            own[name].copy_(param)

    def live_parameters(self):
        """(name, parameter) pairs that receive gradients, in registration order."""
        return [(n, p) for n, p in self.named_parameters()
                if ".adagn.y_shift." not in n and not (".attn_layers." in n and ".norm." in n)]

    def grad_blocks(self):
        """Live parameters grouped by the top-level block whose backward completes them, in completion order
        (the order the flat gradient buffer and the data-parallel buckets follow)."""
        L = self.num_layers
        groups = [("out_layers", self.out_layers)]
        groups += [(f"up_layers.{i}", self.up_layers[i]) for i in range(L - 1, -1, -1)]
        groups += [("middle_layer", self.middle_layer)]
        groups += [(f"down_layers.{k}", self.down_layers[k]) for k in range(L - 1, -1, -1)]
        groups += [("in_layer", self.in_layer)]
        if self.cond_emb is not None:
            groups += [("cond_emb", self.cond_emb)]
        live = {id(p) for _, p in self.live_parameters()}
        # The y_scale projections of every AdaGN are USED until the very end of the backward: their data gradient (the
        # embedding gradient) is one batched launch after the last block (ops.LinearBank.backward_dx).  An optimizer
        # that updates a block's parameters as soon as the block has finished must therefore not see them in that
        # block: they complete -- and are listed -- with the embedding MLP.
        late = {id(p) for a in self._adagns for p in (a.y_scale.weight, a.y_scale.bias)} if self.cond_emb is not None else set()
        out = []
        for name, mod in groups:
            ps = [p for p in mod.parameters() if id(p) in live and id(p) not in late]
            if name == "cond_emb":
                ps += [p for a in self._adagns for p in (a.y_scale.weight, a.y_scale.bias)]
            if ps:
                out.append((name, ps))
        return out

    def alloc_grads(self):
        """-> (flat zero-filled fp32 buffer, {param: strided view}, {block: (start, end)}) in completion order."""
        blocks = self.grad_blocks()
        dev = next(self.parameters()).device
        flat = torch.zeros(sum(p.numel() for _, ps in blocks for p in ps), dtype=torch.float32, device=dev)
        grads, ranges, off = {}, {}, 0
        for name, ps in blocks:
            start = off
            for p in ps:
                grads[p] = flat[off:off + p.numel()].as_strided(p.shape, p.stride())
                off += p.numel()
            ranges[name] = (start, off)
        return flat, grads, ranges

    # ------------------------------------------------------------------ forward
    def _check_input(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channel:
            raise RuntimeError(f"expected input [N, {self.in_channel}, H, W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("U_Net runs on MI355X only: move the module and inputs to 'cuda' "
                               "(there is no CPU fallback; the CPU oracle lives under oracle/)")
        div = 1 << self.num_layers
        if x.shape[2] % div or x.shape[3] % div:
            raise RuntimeError(f"H and W must be multiples of 2**num_layers = {div} (skip connections must align)")

    def modulation_tables(self, t_steps, labels=None):
        """AdaGN modulation of a whole sampling run, computed once: (time_part [T, W], cond_part [n, W] or None, column
        ranges [(c0, c1)] per AdaGN in self._adagns order).  With emb = te_t + ce_i (custom_layers.py:92-98) and y_scale
        linear (:40-44), the modulation of step t for image i is  W (te_t + ce_i) + b = time_part[t] + cond_part[i]:
        time_part[t] = W te_t + b over all T steps of the run, cond_part[i] = W ce_i.  The per-step forward then skips the
        sinusoid, both embedding MLPs and every y_scale projection (six launches of a ~100-launch step at batch 1);
        sdm_step_begin gathers the step's rows.  Same arithmetic as the per-step path up to fp32 summation order."""
        if self.cond_emb is None or not self._adagns:
            return None
        dev = next(self.parameters()).device
        te = self.cond_emb.time_rows(t_steps.to(dev))
        ce = None
        if labels is not None and self.cond_emb.cond_layer is not None:
            ce = self.cond_emb.cond_rows(labels.to(dev))
        cols, off = [], 0
        for a in self._adagns:
            O = a.y_scale.weight.shape[0]
            cols.append((off, off + O))
            off += (O + 3) // 4 * 4                     # float4-aligned column ranges
        A = torch.empty((te.shape[0], off), dtype=torch.float32, device=dev)
        B = None if ce is None else torch.empty((ce.shape[0], off), dtype=torch.float32, device=dev)
        for a, (c0, c1) in zip(self._adagns, cols):
            ops.linear_rows(te, a.y_scale.weight, a.y_scale.bias, out=A[:, c0:c1])
            if B is not None:
                ops.linear_rows(ce, a.y_scale.weight, None, out=B[:, c0:c1])
        return A, B, cols

    def _run(self, ctx, x, t, cond):
        x = x.to(torch.float32).contiguous()
        emb = None
        preset = getattr(self, "_preset_mods", None)
        if preset is not None and not ctx.train and self.cond_emb is not None:
            # a sampler tabulated the modulation of the whole run (modulation_tables) and sdm_step_begin has put this
            # step's rows into `preset`: no embedding work in the step
            emb = preset[0]                              # (only its presence matters to the blocks)
            ctx.emb = emb
            ctx.mods = dict(zip(self._adagns, preset))
        elif self.cond_emb is not None:
            emb = self.cond_emb.run(ctx, t.to(x.device), None if cond is None else cond.to(x.device))
            if ctx.train and emb.shape[0] != x.shape[0]:
                raise RuntimeError("training needs one timestep per image (t of shape [N])")
            ctx.emb = emb
            if self._adagns:             # all y_scale projections of the net in one launch
                views = self._bank.forward([a.y_scale for a in self._adagns], emb, private=ctx.train)
                ctx.mods = dict(zip(self._adagns, views))
                if ctx.train:
                    ctx.bank = self._bank.tables()
                    ctx.dmods = dict(zip(self._adagns, ctx.bank["dviews"]))
        stem = self.in_layer[0].conv_layer.conv
        a = ops.stem_conv(x, stem.taps(), stem.bias, ACT_SILU, save_z=ctx.train,
                          out_dtype=torch.bfloat16 if ctx.bf16 else torch.float32)
        if ctx.train:
            ctx.rec["x"] = x
        a, _ = self.in_layer[1].run(ctx, a)
        skips = []
        for blk in self.down_layers:
            a = blk.run(ctx, a, emb)
            skips.append(a)
        a, _ = self.middle_layer[0].run(ctx, a)
        a, _ = self.middle_layer[1].run(ctx, a)
        for blk in self.up_layers:
            s = skips.pop()
            a = blk.run(ctx, Act(a.t, a.n, a.h, a.w, t2=s.t, z=a.z), emb)     # torch.cat((x, skip), 1) never materialised
        a, _ = self.out_layers[0].run(ctx, a)
        head = self.out_layers[1].conv_layer.conv
        if ctx.train:
            ctx.rec["head_in"] = a
        return ops.head_conv(a, head.taps(), head.bias, ACT_TANH if self.image_recon else ACT_NONE,
                             wt16=head.taps_for(ctx) if ctx.bf16 else None)

    def forward(self, x, t=None, cond=None):
        self._check_input(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            live = self.live_parameters()
            return _UNetFunction.apply(self, x, t, cond, *[p for _, p in live])
        return self._run(RunCtx(False, bf16=self.inference_dtype == torch.bfloat16), x, t, cond)

    def set_inference_dtype(self, dtype):
        """Storage type of activations / contraction weights for no-grad forwards (sampling): torch.float32 (default,
        the parity path) or torch.bfloat16 (fp32 accumulation and statistics; rel-L2 ~ 3e-3 vs fp32).  Training and
        autograd forwards are always fp32.  Additive API: the reference has no such switch."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("inference dtype must be torch.float32 or torch.bfloat16")
        self.inference_dtype = dtype
        return self

    def _edge_slab(self, dev):
        """Scratch for the per-workgroup partial slabs of the stem/head weight gradients (persistent: graph-safe)."""
        need = max(ops.lib.raw("sdm_edge_wgrad_workspace_bytes")(self.out_channel, self.channel_layers[0]),
                   ops.lib.raw("sdm_edge_wgrad_workspace_bytes")(self.in_channel, self.channel_layers[0])) // 4
        slab = getattr(self, "_slab", None)
        if slab is None or slab.numel() < need or slab.device != dev:
            self._slab = slab = torch.empty(need, dtype=torch.float32, device=dev)
        return slab

    # ------------------------------------------------------------------ backward
    def _backward(self, ctx, out, dout, grads):
        """grads: {parameter: ZERO-FILLED gradient buffer with the parameter's layout}, accumulated into."""
        L = self.num_layers
        dout = dout.to(torch.float32).contiguous()
        if self.image_recon:
            dout = ops.tanh_bwd(dout, out)
        if ctx.emb is not None:
            ctx.demb = torch.zeros_like(ctx.emb)
        head = self.out_layers[1].conv_layer.conv
        hin = ctx.rec["head_in"]
        n, H, W = hin.n, hin.h, hin.w
        st = ops._stream()

        bf = ctx.bf16
        if bf:
            # bf16 path: the 3-channel gradient becomes NHWC rows padded to 8 channels, and the head conv's gradients run
            # on the matrix-core kernels like any other conv (weight gradient: M = 3 rows of a 64-row tile; data gradient:
            # K = 8 of a 64-deep K-step -- wasteful per FLOP, but ~50x faster than the one-wave-per-pixel forms at 256x256)
            dy8 = ops.nchw_to_rows_bf16(dout, 8)
            gw = head.grad_taps(grads[head.weight])
            if ops.SIDE.enabled:
                with ops.SIDE.fork(hin.t, dy8):
                    ops.conv_wgrad(dy8[:, :self.out_channel], (H, W), hin, "c3s1", gw, dbias=grads[head.bias])
            else:
                ops.conv_wgrad(dy8[:, :self.out_channel], (H, W), hin, "c3s1", gw, dbias=grads[head.bias])
            if hin.c % 32 == 0 and self.out_channel <= 8 and HEAD_DGRAD_EDGE[0]:
                # the transposed 3 -> C conv as its own matrix-core kernel (csrc/sdm_edge_bf16.hip, data-gradient mode): fp32
                # gradient image and fp32 taps in, silu'(z) applied, bf16 rows out -- 3x faster than K = 8 of a 64-deep K-step
                dz = torch.empty_like(hin.t)
                ops.lib.call("sdm_head_conv_dgrad_bf16", dout.data_ptr(), head.taps().data_ptr(), dz.data_ptr(),
                             hin.z.data_ptr(), n, hin.c, H, W, self.out_channel, st)
            else:
                dz = ops.conv_dgrad(dy8, (H, W), head.taps_t_for(ctx), "c3s1", (H, W), n, dact_z=hin.z)
        else:
            # weight gradient of the 128 -> 3 head conv on the matrix cores: the 3-channel gradient as NHWC rows (padded to 4)
            # through the same kernel as every other conv (3x faster than the one-wave-per-pixel kernel + slab reduction)
            dy4 = ops.nchw_to_rows(dout, (self.out_channel + 3) // 4 * 4)

            def head_wgrad():
                ops.conv_wgrad(dy4[:, :self.out_channel], (H, W), hin, "c3s1", head.grad_taps(grads[head.weight]),
                               dbias=grads[head.bias])
            if ops.SIDE.enabled:       # edge weight gradients ride the side stream too (fork keeps dy4 alive until the join)
                with ops.SIDE.fork(hin.t, dy4):
                    head_wgrad()
            else:
                head_wgrad()
            dz = torch.empty_like(hin.t)
            ops.lib.call("sdm_head_conv_dgrad", dout.data_ptr(), head.taps().data_ptr(), dz.data_ptr(), hin.z.data_ptr(), n,
                         hin.c, H, W, self.out_channel, st)
        dz = self.out_layers[0].backward(ctx, dz, grads)                 # -> dz of the last up-sampler
        user_hook = getattr(self, "_block_done_hook", None)

        def hook(name):            # every gradient of the block is enqueued (weight gradients on ops.SIDE's stream:
            if user_hook is not None:   # a consumer must wait for BOTH streams, see DiffusionTrainer._bucket_closed)
                user_hook(name)
        hook("out_layers")
        pending = {}
        for idx in range(L - 1, -1, -1):                                   # up_layers[idx] handles level k = L-1-idx
            blk = self.up_layers[idx]
            g = blk.backward(ctx, dz, grads)                               # [rows, c_prev + c_skip]
            c_prev = ctx.rec[blk.res_layers[0].conv_block_1][0].c1
            dz = g[:, :c_prev]                                             # silu' of the producer already applied
            pending[L - 1 - idx] = g[:, c_prev:]                           # skip half: finished on the way down
            hook(f"up_layers.{idx}")
        dz = self.middle_layer[1].backward(ctx, dz, grads)
        dz = self.middle_layer[0].backward(ctx, dz, grads, res=pending[L - 1])
        hook("middle_layer")
        for k in range(L - 1, -1, -1):
            dz = self.down_layers[k].backward(ctx, dz, grads, extra=pending[k - 1] if k >= 1 else None)
            hook(f"down_layers.{k}")
        dz = self.in_layer[1].backward(ctx, dz, grads)
        stem = self.in_layer[0].conv_layer.conv
        x = ctx.rec["x"]
        if bf:
            x8 = ops.nchw_to_rows_bf16(x, 8)
            xa = ops.Act(x8[:, :x.shape[1]], x.shape[0], x.shape[2], x.shape[3])
            gw = stem.grad_taps(grads[stem.weight])
            if ops.SIDE.enabled:
                with ops.SIDE.fork(x8, dz):
                    ops.conv_wgrad(dz, (x.shape[2], x.shape[3]), xa, "c3s1", gw, dbias=grads[stem.bias])
            else:
                ops.conv_wgrad(dz, (x.shape[2], x.shape[3]), xa, "c3s1", gw, dbias=grads[stem.bias])
        else:
            x4 = ops.nchw_to_rows(x, (x.shape[1] + 3) // 4 * 4)

            def stem_wgrad():
                ops.conv_wgrad(dz, (x.shape[2], x.shape[3]), ops.Act(x4[:, :x.shape[1]], x.shape[0], x.shape[2], x.shape[3]), "c3s1",
                               stem.grad_taps(grads[stem.weight]), dbias=grads[stem.bias])
            if ops.SIDE.enabled:
                with ops.SIDE.fork(x4, dz):
                    stem_wgrad()
            else:
                stem_wgrad()
        hook("in_layer")
        if self.cond_emb is not None:
            if ctx.dmods is not None:
                self._bank.backward_dx(ctx.bank, ctx.demb)
            self.cond_emb.backward(ctx, ctx.demb, grads)
            hook("cond_emb")
        ops.SIDE.join()

    def set_training_dtype(self, dtype):
        """Storage type of activations, activation gradients and the contraction weight copies in TRAINING forwards /
        backwards: torch.float32 (default, the parity path) or torch.bfloat16 (fp32 master weights, fp32 weight gradients
        and optimizer; every sum fp32; gradients within rel-L2 ~1e-2 of fp32).  The counterpart of the reference's fp16
        autocast (train_diffusion.py:333); additive API."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("training dtype must be torch.float32 or torch.bfloat16")
        self.training_dtype = dtype
        return self

    def loss_and_grads(self, x, t, cond, target, grads=None):
        """One training forward+backward of mean-squared error against ``target`` without going through autograd:
        returns (loss tensor [1], {param: grad}).  Used by the fused trainer and the benchmark."""
        self._check_input(x)
        ctx = RunCtx(True, bf16=self.training_dtype == torch.bfloat16)
        out = self._run(ctx, x, t, cond)
        loss, dout = ops.mse(out, target.contiguous(), want_grad=True)
        if grads is None:
            _, grads, _ = self.alloc_grads()
        self._backward(ctx, out, dout, grads)
        return loss, grads


class _UNetFunction(torch.autograd.Function):
    """The whole U-Net as one autograd node (inputs: the live parameters)."""

    @staticmethod
    def forward(fctx, net, x, t, cond, *params):
        ctx = RunCtx(True, bf16=net.training_dtype == torch.bfloat16)
        out = net._run(ctx, x.detach(), t, cond)
        fctx.net, fctx.run, fctx.out, fctx.params = net, ctx, out, params
        return out

    @staticmethod
    def backward(fctx, dout):
        net = fctx.net
        _, grads, _ = net.alloc_grads()
        net._backward(fctx.run, fctx.out, dout, grads)
        fctx.run = None
        return (None, None, None, None) + tuple(grads[p] for p in fctx.params)
