"""bf16 storage path of the forward (sampling): hand-written bf16 kernels through the C ABI vs fp32 comparators.

The reference has no bf16 mode, so nothing it ships pins these numbers ("parity unpinned" for the bf16 rounding
itself).  What is checked: (a) each bf16 kernel against a torch fp32 computation on the SAME bf16-rounded operands
(only the output rounding and fp32 summation order differ: rel-L2 <= 4e-3), (b) the whole U-Net in bf16 against the
golden fp32 outputs of the reference (rel-L2 <= 1e-2, SURVEY §4's suggested bf16 gate) and (c) a DDIM trajectory.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    torch.cuda.synchronize()
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import unet_oracle
    return unet_oracle


def rnd(x):
    """fp32 tensor rounded through bf16 (what the bf16 kernels see)."""
    return x.to(BF).to(torch.float32)


def torch_conv(x, w, b, kind):
    import torch.nn.functional as F
    if kind == "c3s1":
        return F.conv2d(x, w, b, padding=1)
    if kind == "c3s2":
        return F.conv2d(x, w, b, stride=2, padding=1)
    return F.conv_transpose2d(x, w, b, stride=2, padding=1)


CONV_CASES = [
    # kind, n, cin, cout, h, w
    ("c3s1", 2, 32, 64, 6, 10),
    ("c3s1", 1, 128, 128, 16, 16),
    ("c3s1", 3, 40, 24, 5, 7),          # K, N not multiples of the tile (but of 8)
    ("c3s1", 1, 256, 256, 4, 4),        # small M: split-K path
    ("c3s1", 8, 64, 128, 32, 32),       # 128x128-tile path (M = 8192)
    ("c3s1", 16, 64, 256, 64, 64),      # 256x256-tile path (M = 65536, N = 256: one full round of 256 tiles)
    ("c3s1", 12, 64, 128, 64, 64),      # row-patch kernel as planned (384 tiles of 128x128, 3x3 stride-1)
    ("c3s1", 32, 64, 128, 64, 64),      # ... its opt-in 8-wave 256x128 tile when forced (512 tiles of 256 pixels)
    ("c3s1", 10, 64, 96, 128, 128),     # ... with a ragged N (96 of 128 columns) and two image rows per tile
    ("c3s1", 1, 128, 72, 8, 16),        # ... its edge cases when forced (test below): one tile = a whole 8x16 image,
    ("c3s1", 2, 64, 128, 2, 64),        # a tile = two image rows, ragged N
    ("c3s1", 1, 192, 128, 1, 128),      # a one-row image
    ("c3s2", 2, 32, 64, 8, 12),
    ("c3s2", 1, 64, 64, 7, 9),
    ("ct4", 2, 64, 32, 3, 5),
    ("ct4", 1, 128, 64, 8, 8),
]


@pytest.mark.parametrize("kind,n,cin,cout,h,w", CONV_CASES)
def test_conv_bf16(dev, kind, n, cin, cout, h, w):
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(hash((kind, n, cin, cout, h, w)) % 2**31)
    conv = L.TapConv(cin, cout, kind)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (cin * 9) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    x = torch.randn(n, cin, h, w, generator=g)
    z = torch_conv(rnd(x), rnd(conv.weight.detach().contiguous()), conv.bias.detach(), kind)
    ref = z * torch.sigmoid(z)
    conv = conv.to(dev)
    xr = x.permute(0, 2, 3, 1).reshape(n * h * w, cin).contiguous().to(dev).to(BF)
    ctx = L.RunCtx(False, bf16=True)
    out = ops.conv_fwd(ops.Act(xr, n, h, w), conv.taps_for(ctx), conv.bias, kind, ops.ACT_SILU)
    assert out.t.dtype == BF
    got = out.t.float().reshape(n, out.h, out.w, cout).permute(0, 3, 1, 2)
    assert rel(got, ref) < 4e-3, rel(got, ref)


def test_conv_bf16_concat_and_residual(dev):
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(5)
    n, h, w, c1, c2, cout = 2, 8, 8, 32, 64, 96
    conv = L.TapConv(c1 + c2, cout, "c3s1")
    xa, xb = torch.randn(n, c1, h, w, generator=g), torch.randn(n, c2, h, w, generator=g)
    ref = torch_conv(rnd(torch.cat((xa, xb), 1)), rnd(conv.weight.detach().contiguous()), conv.bias.detach(), "c3s1")
    conv = conv.to(dev)

    def r(x):
        return x.permute(0, 2, 3, 1).reshape(n * h * w, -1).contiguous().to(dev).to(BF)

    ctx = L.RunCtx(False, bf16=True)
    out = ops.conv_fwd(ops.Act(r(xa), n, h, w, t2=r(xb)), conv.taps_for(ctx), conv.bias, "c3s1", ops.ACT_NONE)
    got = out.t.float().reshape(n, h, w, cout).permute(0, 3, 1, 2)
    assert rel(got, ref) < 4e-3
    # linear with residual epilogue
    W, b = torch.randn(48, 64, generator=g) / 8, torch.randn(48, generator=g)
    xs, res = torch.randn(200, 64, generator=g), torch.randn(200, 48, generator=g)
    y = ops.linear_rows(xs.to(dev).to(BF), ops.to_bf16(W.to(dev)), b.to(dev), res=res.to(dev).to(BF))
    assert rel(y.float(), rnd(xs) @ rnd(W).T + b + rnd(res)) < 4e-3


def test_groupnorm_fold_and_affine_bf16(dev):
    from sdm_hip import ops
    g = torch.Generator().manual_seed(9)
    n, h, w, c = 3, 6, 5, 64
    y = torch.randn(n, c, h, w, generator=g) * 1.7 + 0.3
    gamma, beta, mod = torch.randn(c, generator=g), torch.randn(c, generator=g), torch.randn(n, c, generator=g)
    x = torch.randn(n, c, h, w, generator=g)
    yr = rnd(y)
    gn = torch.nn.functional.group_norm(yr, 32, gamma, beta, 1e-5)
    ref = gn * mod[:, :, None, None] + mod[:, :, None, None] + rnd(x)

    def r(t):
        return t.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous().to(dev).to(BF)

    ya = ops.Act(r(y), n, h, w)
    scale, shift, _, _ = ops.gn_affine(ya, gamma.to(dev), beta.to(dev), mod.to(dev).contiguous(), 32)
    out = ops.affine_residual(ya, scale, shift, ops.Act(r(x), n, h, w))
    got = out.t.float().reshape(n, h, w, c).permute(0, 3, 1, 2)
    assert rel(got, ref) < 4e-3, rel(got, ref)
    only = ops.affine_residual(ya, scale, shift, None)      # affine only (next conv's input)
    assert rel(only.t.float().reshape(n, h, w, c).permute(0, 3, 1, 2), ref - rnd(x)) < 4e-3


@pytest.mark.parametrize("n,S,heads,dk", [(2, 36, 2, 16), (1, 256, 1, 64), (3, 100, 1, 128)])
def test_attention_core_bf16(dev, n, S, heads, dk):
    from sdm_hip import ops
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(n * S, heads * 3 * dk, generator=g)
    q4 = rnd(qkv).reshape(n, S, heads, 3 * dk)
    q, k, v = q4[..., :dk], q4[..., dk:2 * dk], q4[..., 2 * dk:]
    attn = torch.einsum("bihd,bjhd->bijh", q, k) * dk ** -0.5
    attn = attn.softmax(dim=1)                                  # over queries (custom_layers.py:147)
    ref = torch.einsum("bijh,bjhd->bihd", attn, v).reshape(n * S, heads * dk)
    got = ops.attention_core_fwd_bf16(qkv.to(dev).to(BF), n, S, heads, dk)
    assert got.dtype == BF
    assert rel(got.float(), ref) < 6e-3, rel(got.float(), ref)


CASES = ["narrow_cond", "narrow_gain", "deep_uncond", "sr_recon", "no_emb"]


@pytest.mark.parametrize("case", CASES)
def test_unet_forward_bf16_vs_golden(dev, O, case):
    """Whole network in bf16 storage against the reference's fp32 outputs (fixtures made by importing the reference)."""
    import json
    from models.U_Net import U_Net
    g = np.load(os.path.join(GOLDEN, f"model_{case}.npz"))
    cfg = json.loads(str(g["cfg"]))
    net = U_Net(**cfg)
    net.load_state_dict(O.seeded_params(cfg, seed=3, gain=float(g["gain"])), strict=True)
    net = net.cuda().eval().set_inference_dtype(BF)
    x, t = T(g["x"]).cuda(), T(g["t"]).cuda()
    cond = T(g["cond"]).cuda() if "cond" in g.files else None
    with torch.no_grad():
        y = net(x, t if cfg["time_dim"] is not None else None, cond)
    assert y.dtype == torch.float32
    r = rel(y, T(g["y"]))
    assert r < 1e-2, r                                          # outer bound: the reference's fp32 output
    # derived gate: the oracle's bf16-storage mode and its jittered twin (tests/bf16_gates.py)
    import bf16_gates as BG
    P = O.seeded_params(cfg, seed=3, gain=float(g["gain"]))
    xx, tt = T(g["x"]), (T(g["t"]) if cfg["time_dim"] is not None else None)
    cc = T(g["cond"]) if "cond" in g.files else None
    with torch.no_grad():
        yb = O.unet_forward(P, cfg, xx, tt, cc, storage=torch.bfloat16)
        ytw = O.unet_forward(P, cfg, xx, tt, cc, storage=O.Bf16Storage().floor_twin(jitter=BG.TWIN_JITTER))
    BG.check_forward(y, yb, T(g["y"]), ytw, what=case)
    net.set_inference_dtype(torch.float32)                      # and back: the fp32 path is untouched
    with torch.no_grad():
        y32 = net(x, t if cfg["time_dim"] is not None else None, cond)
    assert rel(y32, T(g["y"])) < 2e-5


def test_standard_architecture_bf16_vs_oracle(dev, O):
    from models.U_Net import U_Net
    cfg = dict(O.STANDARD_CFG)
    torch.manual_seed(0)
    net = U_Net(**cfg)
    P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
    net = net.cuda().eval().set_inference_dtype(BF)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 3, 64, 64, generator=g)
    t = torch.tensor([753, 12])
    import bf16_gates as BG
    with torch.no_grad():
        ref = O.unet_forward(P, cfg, x, t)
        y = net(x.cuda(), t.cuda())
        yb = O.unet_forward(P, cfg, x, t, storage=torch.bfloat16)
        ytw = O.unet_forward(P, cfg, x, t, storage=O.Bf16Storage().floor_twin(jitter=BG.TWIN_JITTER))
    r = rel(y, ref)
    assert r < 1e-2, r
    BG.check_forward(y, yb, ref, ytw, what="standard 64x64 N=2")


@pytest.mark.parametrize("use_graph", [False, True])
def test_ddim_trajectory_bf16_close_to_fp32(dev, O, use_graph):
    """DDIM steps with the bf16 network stay close to the fp32 trajectory of the same seeded net (sampler state and
    arithmetic are fp32 in both).  The net is random-init, not a trained denoiser: the trajectory amplifies
    perturbations, hence the loose bound."""
    import degraders
    import diffusion_sampling_algorithms as S
    from models.U_Net import U_Net
    narrow = dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3, num_layers=2,
                  attn_layers=[1], num_heads=2, dim_per_head=16, min_channel=32, max_channel=64)
    cfg = O.full_cfg(narrow)
    net = U_Net(**cfg)
    net.load_state_dict(O.seeded_params(cfg, seed=3, gain=1.0), strict=True)
    net = net.cuda().eval()
    sched = degraders.CosineNoiseDegradation(1000)
    g = torch.Generator().manual_seed(4)
    noise = torch.randn(2, 3, 16, 16, generator=g).cuda()
    lab = torch.tensor([0.0, 1.0, 0.0]).cuda()
    quiet = lambda *a, **k: None
    x32 = S.ddim_sampling(net, sched, noise.clone(), 1, 1000, None, lab, 100, "cuda", quiet, use_graph=use_graph)
    net.set_inference_dtype(BF)
    x16 = S.ddim_sampling(net, sched, noise.clone(), 1, 1000, None, lab, 100, "cuda", quiet, use_graph=use_graph)
    assert torch.isfinite(x16).all()
    assert rel(x16, x32) < 5e-2, rel(x16, x32)


def test_bf16_weights_follow_the_optimizer(dev, O):
    """The cached bf16 weight copies are rebuilt after a fused-Adam step (which writes weights through raw pointers)."""
    import degraders
    from models.U_Net import U_Net
    from sdm_hip.trainer import DiffusionTrainer
    narrow = dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=None, num_layers=2,
                  attn_layers=[1], num_heads=1, dim_per_head=None, min_channel=32, max_channel=64)
    cfg = O.full_cfg(narrow)
    net = U_Net(**cfg)
    net.load_state_dict(O.seeded_params(cfg, seed=3, gain=1.5), strict=True)
    net = net.cuda().set_inference_dtype(BF)
    x = torch.randn(2, 3, 16, 16, device="cuda")
    t = torch.tensor([500, 20], device="cuda")
    net.eval()
    with torch.no_grad():
        y0 = net(x, t).clone()
    tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-3, seed=1)
    net.train()
    tr.step(torch.randn(2, 3, 16, 16, device="cuda"))
    torch.cuda.synchronize()
    net.eval()
    with torch.no_grad():
        y1 = net(x, t)
        net.set_inference_dtype(torch.float32)
        y1_32 = net(x, t)
    assert rel(y1, y0) > 2e-3                # the step moved the weights and the bf16 forward saw it
    assert rel(y1, y1_32) < 1e-2


@pytest.mark.parametrize("n,c,h,w", [(1, 128, 4, 4), (2, 128, 16, 16), (2, 256, 24, 24)])
def test_residual_block_fused_tail_bf16(dev, n, c, h, w):
    """bf16 rows: the fused conv-block tail (sdm_gn_apply_fused_bf16, with and without a deferred split-K epilogue) against
    the unfused bf16 path (same roundings: equal up to summation order) and the fp32 path (bf16 storage noise)."""
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(n * 100 + c + h)
    blk = L.ResidualBlock(c, c, emb_dim=64)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.3))
        blk.conv_block_1.adagn.group_norm.weight.add_(1.0)
        blk.conv_block_2.adagn.group_norm.weight.add_(1.0)
    blk = blk.to(dev)
    x = torch.randn(n * h * w, c, generator=g).cuda()
    emb = torch.randn(n, 64, generator=g).cuda()
    with torch.no_grad():
        ref32 = blk.run(L.RunCtx(False), ops.Act(x, n, h, w), emb).t
        L.FUSE_GN_BF16[0] = True                  # opt-in on bf16 rows (measured neutral for the sampling step)
        fused = blk.run(L.RunCtx(False, bf16=True), ops.Act(x.to(torch.bfloat16), n, h, w), emb).t
        L.FUSE_GN_BF16[0] = False
        plain = blk.run(L.RunCtx(False, bf16=True), ops.Act(x.to(torch.bfloat16), n, h, w), emb).t
    torch.cuda.synchronize()
    assert fused.dtype == torch.bfloat16
    assert rel(fused, plain) < 4e-3, rel(fused, plain)
    assert rel(fused, ref32) < 1e-2, rel(fused, ref32)
    assert ops._all_zero(ops.WS)


@pytest.mark.parametrize("n,c,h,w,expect", [(10, 128, 64, 64, True), (33, 128, 64, 64, True), (24, 256, 32, 32, True), (16, 256, 64, 64, True), (2, 128, 16, 16, None),
                                            (3, 64, 8, 16, None), (1, 128, 12, 12, False)])
def test_groupnorm_statistics_from_the_conv_epilogue(dev, n, c, h, w, expect):
    """sdm_igemm_bf16 with gn_sums: per-(image, channel) sum / sum of squares of the stored outputs from the conv's epilogue
    and sdm_gn_finalize, against the separate statistics pass (sdm_gn_affine_bf16) on the same conv output; shapes where the
    library emits them (large unsplit convs, tiles inside one image), where it may split K instead (small ones: either way
    the tables agree) and where it must decline (12x12 = 144 pixels is no multiple of the tile rows)."""
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(n + c + h * w)
    conv = L.TapConv(c, c, "c3s1").to(dev)
    ctx = L.RunCtx(True, bf16=True)
    x = ops.Act(torch.randn(n * h * w, c, generator=g).cuda().to(torch.bfloat16), n, h, w)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    mod = torch.randn(n, c, generator=g).cuda()
    ops.CONV_GN_STATS[0] = True
    y = ops.conv_fwd(x, conv.taps_for(ctx), conv.bias, "c3s1", ops.ACT_SILU, save_z=True, gn_stats=True)
    emitted = y.stats is not None
    assert expect is None or emitted == expect
    s1, h1, m1, r1 = ops.gn_affine(y, gamma, beta, mod, want_stats=True)
    assert y.stats is None
    y2 = ops.Act(y.t, n, h, w)
    s0, h0, m0, r0 = ops.gn_affine(y2, gamma, beta, mod, want_stats=True)
    for a, b in ((s1, s0), (h1, h0), (m1, m0), (r1, r0)):
        assert rel(a, b) < 2e-5, rel(a, b)
    assert ops._all_zero(ops.GNSUMS)


@pytest.mark.parametrize("force", ["2,1", "2,1+legacy", "3,1", "2,1+pingpong"])
def test_large_tile_configurations_forced(force):
    """The 256x256 / 8-wave tiles of sdm_igemm_bf16 (the phase-interleaved kernel of sdm_igemm8.hip: "2,1"; the
    one-barrier-per-K-step tiles it replaced, kept for A/B: "+legacy" = SDM_BF16_8PH=0, "3,1", "+pingpong") and
    sdm_wgrad_bf16 (planned only for launches that fill the chip with
    them) forced onto every shape that can take them (SDM_BF16_FORCE; read once per process -> subprocess): ragged M / N far below the tile, concat
    sources, residual epilogues, fp32 scores, GroupNorm statistics, data gradients and the golden-pinned U-Net forward."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    pp = force.endswith("+pingpong")       # the opt-in ping-pong schedule of the 8-wave tile (SDM_BF16_PP=1), same cases
    legacy = pp or force.endswith("+legacy")
    force = force.split("+")[0]
    env = dict(os.environ, SDM_BF16_FORCE=force, SDM_WGRAD_BF16_FORCE="256,2")     # weight gradients: 256x256 tile, 2 slices
    if pp:
        env["SDM_BF16_PP"] = "1"
    if legacy:
        env["SDM_BF16_8PH"] = "0"
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_bf16.py"),
                        os.path.join(here, "test_gpu_bf16_train.py"), "-q", "-x", "-k",
                        "test_conv_bf16 or concat_and_residual or attention_core or statistics_from or fwd_dgrad_wgrad or "
                        "wgrad_bf16_concat or token_linear or forward_bf16_vs_golden or grads_bf16_vs_reference_golden"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_row_patch_kernel_forced():
    """conv3_row_bf16_kernel (3x3 stride-1 convs: one activation patch per kernel row serves its three column taps; planned
    where the 128x128 tiles would run) forced onto every eligible shape of the bf16 suites (SDM_BF16_ROW=1; subprocess):
    single-tile images, tiles of several rows, one-row images, ragged channel counts, concat sources, residual / SiLU'
    epilogues of the data gradient, GroupNorm statistics, and the golden-pinned forward / gradients of whole networks."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, SDM_BF16_ROW="1", SDM_BF16_ROW256="3")     # (and the opt-in 256x128 tile, deep weight ring, where eligible)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_bf16.py"),
                        os.path.join(here, "test_gpu_bf16_train.py"), "-q", "-x", "-k",
                        "test_conv_bf16 or concat_and_residual or statistics_from or fwd_dgrad_wgrad or "
                        "forward_bf16_vs_golden or grads_bf16_vs_reference_golden or fused_tail or standard_architecture"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize("n,S,heads,dk", [(2, 1024, 1, 512), (1, 512, 2, 64), (1, 4096, 1, 128), (3, 264, 1, 64)])
def test_two_pass_column_softmax_attention(dev, n, S, heads, dk):
    """bf16 attention core for S > 256 without the S x S fp32 scores in HBM (custom_layers.py:144-150): pass 1 of the scores
    GEMM leaves per-M-tile column statistics, pass 2 writes exp(s - m_j) / l_j as bf16.  Against a float64 computation on
    the same bf16 rows, against the materialised-scores path it replaces, for one / several heads, tile-ragged S, and for
    logits of 1e13 (rounded products must be identical in both passes: no fma contraction)."""
    from sdm_hip import ops
    g = torch.Generator().manual_seed(S + dk)
    qkv = torch.randn(n * S, heads * 3 * dk, generator=g)
    v = qkv.view(n, S, heads, 3, dk)
    gain = 6.0 / dk ** 0.25
    v[:, :, :, 0] *= gain
    v[:, :, :, 1] *= gain
    x = qkv.to(dev).to(torch.bfloat16)
    xr = x.double().view(n, S, heads, 3, dk).cpu()
    q, k, vv = xr[:, :, :, 0], xr[:, :, :, 1], xr[:, :, :, 2]
    attn = (torch.einsum("nihd,njhd->nijh", q, k) * dk ** -0.5).softmax(dim=1)
    want = torch.einsum("nijh,njhd->nihd", attn, vv).reshape(n * S, heads * dk)
    assert ops.TWO_PASS_ATTN[0] and not ops._small_attn_ok(x, n, heads, S, dk)
    res, P = ops.attention_core_fwd_bf16(x, n, S, heads, dk, keep_p=True)
    Pw = attn.permute(0, 3, 1, 2).reshape(n * heads, S, S)
    assert rel(P[:, :, :S].float(), Pw) < 6e-3, rel(P[:, :, :S].float(), Pw)
    assert rel(res.float(), want) < 8e-3, rel(res.float(), want)
    ops.TWO_PASS_ATTN[0] = False
    try:
        res2, P2 = ops.attention_core_fwd_bf16(x, n, S, heads, dk, keep_p=True)
    finally:
        ops.TWO_PASS_ATTN[0] = True
    assert P2.shape == P.shape and rel(P.float(), P2.float()) < 6e-3 and rel(res.float(), res2.float()) < 8e-3
    big = (torch.randn(n * S, heads * 3 * dk, generator=g) * 3e6).to(dev).to(torch.bfloat16)
    r1, P1 = ops.attention_core_fwd_bf16(big, n, S, heads, dk, keep_p=True)
    assert bool(torch.isfinite(r1.float()).all()) and bool(torch.isfinite(P1.float()).all())
    assert float((P1[:, :, :S].float().sum(1) - 1).abs().max()) < 2e-2


HEAD_CASES = [
    # n, cin, cout, h, w, act
    (2, 128, 3, 19, 37, "none"),      # ragged strips (37 = 2 x 14 + 9), one short segment
    (1, 64, 1, 8, 8, "tanh"),         # one strip, 64 channels
    (3, 128, 4, 64, 64, "none"),      # the benchmark geometry (5 strips of 14)
    (1, 128, 16, 33, 15, "tanh"),     # full 16 output rows of the matrix tile, 33 = 4 segments of 8 + 1
    (2, 64, 3, 1, 70, "none"),        # a one-row image
]


@pytest.mark.parametrize("n,cin,cout,h,w,act", HEAD_CASES)
def test_head_conv_strip_kernel(dev, n, cin, cout, h, w, act):
    """sdm_head_conv_mfma_bf16 (csrc/sdm_edge_bf16.hip) vs torch's fp32 conv on the SAME bf16-rounded rows and weights: only
    the fp32 summation order differs (the result itself is fp32), gate 2e-5; the sdm_igemm_bf16 route it replaces must agree."""
    import torch.nn.functional as F
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(hash((n, cin, cout, h, w)) % 2**31)
    conv = L.TapConv(cin, cout, "c3s1")
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (cin * 9) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    x = torch.randn(n, cin, h, w, generator=g)
    ref = F.conv2d(rnd(x).double(), rnd(conv.weight.detach()).double(), conv.bias.detach().double(), padding=1)
    a = ops.ACT_NONE
    if act == "tanh":
        ref, a = torch.tanh(ref), ops.ACT_TANH
    conv = conv.to(dev)
    xr = x.permute(0, 2, 3, 1).reshape(n * h * w, cin).contiguous().to(dev).to(BF)
    ctx = L.RunCtx(False, bf16=True)
    assert ops.HEAD_STRIP[0]
    w16 = conv.taps_for(ctx)                                            # (the bf16 copy of the taps is made here)
    launches = ops.lib.raw("sdm_launch_count")()
    got = ops.head_conv(ops.Act(xr, n, h, w), conv.taps(), conv.bias, a, wt16=w16)
    assert ops.lib.raw("sdm_launch_count")() - launches == 1          # one kernel: no separate layout pass
    assert got.shape == (n, cout, h, w) and got.dtype == torch.float32
    assert rel(got, ref) < 2e-5, rel(got, ref)
    ops.HEAD_STRIP[0] = False
    try:
        old = ops.head_conv(ops.Act(xr, n, h, w), conv.taps(), conv.bias, a, wt16=conv.taps_for(ctx))
    finally:
        ops.HEAD_STRIP[0] = True
    assert rel(got, old) < 2e-5


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 3, 128, 19, 37), (1, 1, 32, 8, 8), (3, 3, 64, 32, 32), (1, 4, 256, 5, 7),
                                            (1, 2, 96, 16, 16), (2, 6, 128, 24, 24), (1, 8, 64, 9, 11)])
def test_stem_conv_matrix_core_kernel(dev, n, cin, cout, h, w, monkeypatch):
    """sdm_stem_conv_bf16's fp32 matrix-core form vs torch's conv: same fp32 operands, bf16 output rounding (1 ulp = 2^-8
    relative on an element whose fp32 value sits on a rounding boundary; rel-L2 gate 4e-3 as for every bf16 kernel)."""
    import torch.nn.functional as F
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(hash((n, cin, cout, h, w)) % 2**31)
    conv = L.TapConv(cin, cout, "c3s1")
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (cin * 9) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    x = torch.randn(n, cin, h, w, generator=g)
    z = F.conv2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=1)
    ref = z * torch.sigmoid(z)
    conv = conv.to(dev)
    out = ops.stem_conv(x.to(dev), conv.taps(), conv.bias, ops.ACT_SILU, save_z=True, out_dtype=BF)
    got = out.t.float().reshape(n, h, w, cout).permute(0, 3, 1, 2)
    gz = out.z.float().reshape(n, h, w, cout).permute(0, 3, 1, 2)
    assert rel(got, ref) < 4e-3 and rel(gz, z) < 4e-3, (rel(got, ref), rel(gz, z))
    # element-wise: never more than one bf16 ulp away from the exactly rounded pre-activation
    torch.cuda.synchronize()
    zc = z.float().to(BF).float()
    assert float(((gz.cpu() - zc).abs() / (zc.abs() + 1e-3)).max()) < 2.0 ** -7


@pytest.mark.parametrize("n,c,cout,h,w,with_z", [(2, 128, 3, 19, 37, True), (1, 64, 1, 8, 8, False), (2, 256, 4, 16, 16, True),
                                                 (1, 96, 3, 7, 5, True)])
def test_head_conv_data_gradient_matrix_core_kernel(dev, n, c, cout, h, w, with_z):
    """sdm_head_conv_dgrad_bf16 (data-gradient mode of the stem kernel in csrc/sdm_edge_bf16.hip) vs torch's transposed conv
    in float64 on the same fp32 gradient image and fp32 taps, times silu'(z) of the bf16 pre-activation: only the output is
    rounded to bf16 (rel-L2 gate 4e-3, element-wise within one bf16 ulp of the rounded exact value)."""
    import torch.nn.functional as F
    from models import custom_layers as L
    from sdm_hip import ops
    g = torch.Generator().manual_seed(hash((n, c, cout, h, w)) % 2**31)
    conv = L.TapConv(c, cout, "c3s1")
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (cout * 9) ** 0.5)
    dy = torch.randn(n, cout, h, w, generator=g)
    z = torch.randn(n, c, h, w, generator=g).to(BF)
    ref = F.conv_transpose2d(dy.double(), conv.weight.detach().double(), padding=1)
    if with_z:
        zd = z.double()
        sg = torch.sigmoid(zd)
        ref = ref * (sg * (1 + zd * (1 - sg)))
    conv = conv.to(dev)
    zr = z.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous().to(dev)
    dx = torch.empty((n * h * w, c), dtype=BF, device=dev)
    ops.lib.call("sdm_head_conv_dgrad_bf16", dy.to(dev).data_ptr(), conv.taps().data_ptr(), dx.data_ptr(),
                 zr.data_ptr() if with_z else None, n, c, h, w, cout, ops._stream())
    got = dx.float().reshape(n, h, w, c).permute(0, 3, 1, 2)
    assert rel(got, ref) < 4e-3, rel(got, ref)
    rc = ref.float().to(BF).float()
    assert float(((got.cpu() - rc).abs() / (rc.abs() + 1e-3)).max()) < 2.0 ** -6     # (sigmoid by v_rcp: 1 ulp of fp32 in the factor)
