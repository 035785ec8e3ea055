"""The CPU oracle (oracle/unet_oracle.py) against vectors produced by the real
reference (tests/golden/make_golden.py).  Pure CPU; runs in the build container
and on the GPU box alike (never reads /root/reference)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["narrow_cond", "narrow_gain", "deep_uncond", "sr_recon", "no_emb"]
RTOL, ATOL = 1e-3, 1e-4     # north_star gate; the oracle is far inside it (asserted tighter below)


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference(case):
    g = load(f"model_{case}.npz")
    cfg = json.loads(str(g["cfg"]))
    P = O.seeded_params(cfg, seed=3, gain=float(g["gain"]))
    x, t = T(g["x"]), T(g["t"])
    cond = T(g["cond"]) if "cond" in g else None
    with torch.no_grad():
        y = O.unet_forward(P, cfg, x, t if cfg["time_dim"] is not None else None, cond)
    ref = T(g["y"])
    assert torch.allclose(y, ref, rtol=1e-4, atol=2e-6), float((y - ref).abs().max())
    rel = float((y - ref).norm() / ref.norm())
    assert rel < 1e-5
    if cfg["time_dim"] is not None:   # batch-1 embedding broadcast
        with torch.no_grad():
            yb = O.unet_forward(P, cfg, x, t[:1], cond[0] if cond is not None else None)
        assert torch.allclose(yb, T(g["y_b1"]), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_grads_match_reference(case):
    g = load(f"model_{case}.npz")
    cfg = json.loads(str(g["cfg"]))
    P = {k: v.requires_grad_(True) for k, v in O.seeded_params(cfg, seed=3, gain=float(g["gain"])).items()}
    x, t = T(g["x"]), T(g["t"])
    cond = T(g["cond"]) if "cond" in g else None
    y = O.unet_forward(P, cfg, x, t if cfg["time_dim"] is not None else None, cond)
    loss = torch.nn.functional.mse_loss(y, T(g["target"]))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    dead = json.loads(str(g["dead"]))
    assert dead == O.dead_param_names(cfg)
    for k, p in P.items():
        if k in dead:
            assert p.grad is None
            continue
        gr = p.grad.double().flatten()
        ref = T(g[f"gval/{k}"]).double()
        got = gr[T(g[f"gidx/{k}"])]
        scale = float(g[f"gstat/{k}"][2]) / max(1.0, gr.numel() ** 0.5)
        assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-2 * scale + 1e-9, k
        assert abs(float(gr.norm()) - float(g[f"gstat/{k}"][2])) <= 1e-4 * float(g[f"gstat/{k}"][2]) + 1e-9, k


def _params(b, prefix):
    return {k[len(prefix) + 3:]: T(b[k]) for k in b.files if k.startswith(prefix + "/p/")}


def test_blocks_match_reference():
    b = load("blocks.npz")
    # attention with large logits pins the softmax axis and the qkv chunk layout
    for tag, heads, dk in (("attn_h2", 2, 16), ("attn_h1", 1, None)):
        P = {"a." + k: v for k, v in _params(b, tag).items()}
        x = T(b[f"{tag}/x"]).requires_grad_(True)
        y = O.attention(P, "a", x, heads, dk)
        ref = T(b[f"{tag}/y"])
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4), float((y - ref).abs().max())
        y.backward(T(b[f"{tag}/dy"]))
        assert torch.allclose(x.grad, T(b[f"{tag}/dx"]), rtol=1e-3, atol=1e-3)
        # wrong axis must be far away (sanity of the fixture)
        xs = x.detach()
        N, C, H, W = xs.shape
    P = {"g." + k: v for k, v in _params(b, "adagn").items()}
    y = O.adagn(P, "g", T(b["adagn/x"]), T(b["adagn/emb"]))
    assert torch.allclose(y, T(b["adagn/y"]), rtol=1e-4, atol=1e-5)
    y1 = O.adagn(P, "g", T(b["adagn/x"]), T(b["adagn/emb"])[:1])
    assert torch.allclose(y1, T(b["adagn/y_b1"]), rtol=1e-4, atol=1e-5)
    P = {"c." + k: v for k, v in _params(b, "convblock").items()}
    y = O.conv_block(P, "c", T(b["convblock/x"]), T(b["convblock/emb"]))
    assert torch.allclose(y, T(b["convblock/y"]), rtol=1e-4, atol=1e-5)
    P = {"r." + k: v for k, v in _params(b, "resblock").items()}
    y = O.res_block(P, "r", T(b["resblock/x"]), T(b["resblock/emb"]))
    assert torch.allclose(y, T(b["resblock/y"]), rtol=1e-4, atol=1e-5)
    for tag, cd in (("emb_t", None), ("emb_tc", 5)):
        P = {"cond_emb." + k: v for k, v in _params(b, tag).items()}
        cfg = dict(time_dim=32, cond_dim=cd)
        y = O.cond_embedding(P, cfg, T(b[f"{tag}/t"]), T(b[f"{tag}/cond"]) if cd else None)
        assert torch.allclose(y, T(b[f"{tag}/y"]), rtol=1e-4, atol=1e-5)


def test_schedules_and_qsample():
    s = load("schedules_samplers.npz")
    ts = T(s["sched/ts"])
    lin, cos = O.LinearSchedule(5e-3, 9e-3, 1000), O.CosineSchedule(1000)
    for tag, sch in (("lin", lin), ("cos", cos)):
        got = torch.stack(sch.params(ts))
        assert torch.allclose(got, T(s[f"sched/{tag}"]), rtol=1e-6, atol=1e-12), tag
        q = O.q_sample(sch, T(s["q/x0"]), T(s["q/t"]), T(s["q/eps"]))
        assert torch.allclose(q, T(s[f"q/{tag}"]), rtol=1e-6, atol=1e-7)
    # SURVEY a15 known values
    b, a, ab = cos.params(torch.tensor([1, 500, 1000]))
    assert abs(float(ab[0]) - 0.99995875) < 1e-6 and abs(float(ab[1]) - 0.49384364) < 1e-6
    assert abs(float(b[1]) - 0.00314564) < 1e-7 and float(b[2]) == pytest.approx(0.999)


def test_sampler_trajectories():
    s = load("schedules_samplers.npz")
    cfg = O.full_cfg(dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3,
                          num_layers=2, attn_layers=[1], num_heads=2, dim_per_head=16,
                          min_channel=32, max_channel=64))
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    xT, labels = T(s["samp/xT"]), T(s["samp/labels"])
    lin, cos = O.LinearSchedule(5e-3, 9e-3, 1000), O.CosineSchedule(1000)
    for tag, sch in (("lin", lin), ("cos", cos)):
        z = T(s[f"samp/ddpm_{tag}_z"])
        out, _ = O.ddpm_sample(P, cfg, sch, xT, list(z) + [None], 1, 6, labels)
        assert torch.allclose(out, T(s[f"samp/ddpm_{tag}_out"]), rtol=1e-4, atol=1e-5), tag
        out, _ = O.ddim_sample(P, cfg, sch, xT, 1, 1000, 250, labels)
        ref = T(s[f"samp/ddim_{tag}_out"])
        # cosine: first step divides by sqrt(alpha_bar_T) ~ 4e-8 -> relative comparison only
        assert float((out - ref).norm() / ref.norm()) < 1e-3, tag
    out, _ = O.ddim_sample(P, cfg, lin, xT, 3, 13, 5, labels)
    assert torch.allclose(out, T(s["samp/ddim_min3_out"]), rtol=1e-4, atol=1e-5)
    assert O.ddim_steps(1, 1000, 20)[:2] == [1000, 980] and len(O.ddim_steps(1, 1000, 20)) == 51


def test_training_steps():
    g = load("train_narrow.npz")
    cfg = O.full_cfg(dict(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3,
                          num_layers=2, attn_layers=[1], num_heads=2, dim_per_head=16,
                          min_channel=32, max_channel=64))
    tr = O.OracleTrainer(O.seeded_params(cfg, seed=3, gain=1.8), cfg, O.CosineSchedule(1000), lr=2e-3)
    for i in range(3):
        loss = tr.step(T(g[f"x0_{i}"]), T(g[f"t_{i}"]), T(g[f"eps_{i}"]), T(g[f"lab_{i}"]))
        assert abs(loss - float(g["losses"][i])) < 2e-5
    for k, v in tr.P.items():
        got = v.detach().double().flatten()[T(g[f"pidx/{k}"])]
        assert torch.allclose(got, T(g[f"pval/{k}"]).double(), rtol=1e-3, atol=2e-5), k


def test_sr_path():
    """Area interpolation, the cold sampler and the SR training step against the reference's SR goldens."""
    g = load("sr_path.npz")
    cfg = O.full_cfg(dict(num_resnet_blocks=1, in_channel=6, out_channel=3, time_dim=32, cond_dim=None, num_layers=2,
                          attn_layers=[0, 1], num_heads=1, dim_per_head=None, min_channel=32, max_channel=32,
                          image_recon=True))
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    lin = O.LinearSchedule(5e-3, 9e-3, 1000)
    noise, up = T(g["samp/noise"]), O.area_resize(T(g["samp/lr_img"]), (8, 8))
    assert torch.allclose(up, T(g["samp/up"]))
    cond = O.q_sample(lin, up, torch.tensor([250]), noise)
    assert torch.allclose(cond, T(g["samp/cond"]), rtol=1e-6, atol=1e-6)
    x0 = O.cold_sample(P, cfg, lin, noise.clone(), noise, cond, 1, 1000, 250)
    assert torch.allclose(x0, T(g["samp/x0"]), rtol=1e-4, atol=1e-5)
    tr = O.OracleTrainer(P, cfg, lin, lr=2e-3)
    for i in range(2):
        x_in, target = O.sr_batch(lin, T(g[f"train/x0_{i}"]), T(g[f"train/t_{i}"]), T(g[f"train/eps_{i}"]), 4, 8, 250)
        tr.opt.zero_grad()
        loss = torch.nn.functional.mse_loss(O.unet_forward(tr.P, cfg, x_in, T(g[f"train/t_{i}"])), target)
        loss.backward()
        tr.opt.step()
        assert abs(float(loss) - float(g["train/losses"][i])) < 2e-5
    for k, v in tr.P.items():
        got = v.detach().double().flatten()[T(g[f"pidx/{k}"])]
        assert torch.allclose(got, T(g[f"pval/{k}"]).double(), rtol=1e-3, atol=2e-5), k


def test_cold_and_doodle_objectives():
    """x0-target (cold diffusion) and image-conditioned (doodle) step bodies and the doodle DDIM run, against the reference
    (train_noise_cold_diffusion.py:307-348, train_doodle_diffusion.py:283-320,398-410)."""
    g = load("cold_doodle.npz")
    cos = O.CosineSchedule(1000)
    cfg = O.full_cfg(json.loads(str(g["cold/cfg"])))
    tr = O.OracleTrainer(O.seeded_params(cfg, seed=3, gain=1.8), cfg, cos, lr=2e-3)
    for i in range(2):
        loss = tr.step(T(g[f"cold/x0_{i}"]), T(g[f"cold/t_{i}"]), T(g[f"cold/eps_{i}"]), T(g[f"cold/lab_{i}"]), target="x0")
        assert abs(loss - float(g["cold/losses"][i])) < 2e-5
    for k, v in tr.P.items():
        got = v.detach().double().flatten()[T(g[f"cold/pidx/{k}"])]
        assert torch.allclose(got, T(g[f"cold/pval/{k}"]).double(), rtol=1e-3, atol=2e-5), k
    cfg = O.full_cfg(json.loads(str(g["doodle/cfg"])))
    P = O.seeded_params(cfg, seed=3, gain=1.8)
    out, _ = O.ddim_sample(P, cfg, cos, T(g["doodle/samp_xT"]), 1, 1000, 250, cond_img=T(g["doodle/samp_lbl"]))
    assert torch.allclose(out, T(g["doodle/samp_out"]), rtol=1e-4, atol=1e-5)
    tr = O.OracleTrainer(P, cfg, cos, lr=2e-3)
    for i in range(2):
        loss = tr.step(T(g[f"doodle/x0_{i}"]), T(g[f"doodle/t_{i}"]), T(g[f"doodle/eps_{i}"]), None,
                       cond_img=T(g[f"doodle/lbl_{i}"]))
        assert abs(loss - float(g["doodle/losses"][i])) < 2e-5
    for k, v in tr.P.items():
        got = v.detach().double().flatten()[T(g[f"doodle/pidx/{k}"])]
        assert torch.allclose(got, T(g[f"doodle/pval/{k}"]).double(), rtol=1e-3, atol=2e-5), k


def test_reference_written_checkpoint_loads_strictly():
    """tests/golden/ref_checkpoint/: a checkpoint torch.save'd by the REAL reference (model + Adam state), its config
    checkpoint and the export_models.py-shaped config.json.  Loaded with weights_only=True; the keys match the drop-in
    U_Net strictly (dead parameters included) and the oracle reproduces the reference's forward from those weights."""
    from models.U_Net import U_Net
    from generate_images_diffusion import _UNET_KEYS
    d = os.path.join(GOLDEN, "ref_checkpoint")
    ck = torch.load(os.path.join(d, "diffusion_1.pt"), map_location="cpu", weights_only=True)
    cc = torch.load(os.path.join(d, "config_1.pt"), map_location="cpu", weights_only=True)
    assert set(ck) == {"model", "optimizer"} and cc == {"starting_epoch": 0, "global_steps": 1}
    with open(os.path.join(d, "config.json")) as f:
        md = json.load(f)["models"][0]
    kw = {k: md[v] for k, v in _UNET_KEYS.items()}
    net = U_Net(**kw)
    net.load_state_dict(ck["model"], strict=True)
    assert list(net.state_dict().keys()) == list(ck["model"].keys())
    # optimizer state: one entry per LIVE parameter (the 40-odd dead ones never get a gradient), indexed like parameters()
    live = {i for i, (n, _) in enumerate(net.named_parameters())
            if ".adagn.y_shift." not in n and not (".attn_layers." in n and ".norm." in n)}
    assert set(ck["optimizer"]["state"].keys()) == live
    e = np.load(os.path.join(d, "expected.npz"))
    cfg = O.full_cfg(kw)
    with torch.no_grad():
        y = O.unet_forward({k: v for k, v in ck["model"].items()}, cfg, T(e["fwd_x"]), T(e["fwd_t"]), T(e["fwd_lab"]))
    assert torch.allclose(y, T(e["fwd_y"]), rtol=1e-4, atol=2e-6)
    # export_models.py:94-97: BASE entries always carry beta_1 / beta_T; the LINEAR-schedule export of the same checkpoint
    assert md["beta_1"] == 5e-3 and md["beta_T"] == 9e-3
    with open(os.path.join(d, "config_linear.json")) as f:
        ml = json.load(f)["models"][0]
    assert ml["noise_scheduler"] == "LINEAR" and ml["model_name"] == md["model_name"]
    with torch.no_grad():
        out, _ = O.ddim_sample({k: v for k, v in ck["model"].items()}, cfg, O.LinearSchedule(ml["beta_1"], ml["beta_T"], 1000),
                               T(e["lin_xT"]), 1, 1000, 250, labels=torch.tensor([1.0, 0.0]))
    assert torch.allclose(out, T(e["lin_ddim_out"]), rtol=1e-4, atol=1e-5)


def _write_dataset_files(tmp_path, g):
    """The fixture's decoded bytes as lossless PNG files + the reference's TinyDB JSON layout."""
    from PIL import Image
    n = g["bgr_u8"].shape[0]
    for i in range(n):
        Image.fromarray(np.ascontiguousarray(g["bgr_u8"][i][:, :, ::-1])).save(tmp_path / f"im{i}.png")     # BGR -> RGB file
        Image.fromarray(np.ascontiguousarray(g["lab_u8"][i][:, :, ::-1])).save(tmp_path / f"lab{i}.png")
    data = {str(i + 1): {"filename": f"im{i}.png", "doodle": f"lab{i}.png", "smiling": int(g["smiling"][i]),
                         "hat": float(g["hat"][i])} for i in range(n)}
    (tmp_path / "cond.json").write_text(json.dumps({"_default": {}, "Data": data, "Labels": {"1": {"labels": ["smiling", "hat"]}}}))
    (tmp_path / "doodle.json").write_text(json.dumps({"_default": {}, "Data": data, "Labels": {"1": {"labels": ["doodle"]}}}))
    return n


def test_datasets_match_the_reference_datasets_bit_for_bit(tmp_path):
    """SURVEY §8f rank 2 pin (tests/golden/datasets.npz, make_golden.py dataset_case): tensors returned by the REFERENCE's
    ImageDataset / ConditionalImgDataset / DoodleImgDataset.__getitem__ (their normalisation, conversions, label handling
    and initial random.shuffle; cv2.imread / TinyDB stubbed as byte / record sources) == what the build's PIL + json
    datasets return for the same decoded bytes, bit for bit, in the same (seeded) record order."""
    import random
    from custom_dataset.img_dataset import open_dataset, open_pair_dataset
    g = load("datasets.npz")
    n = _write_dataset_files(tmp_path, g)
    ds = open_dataset(str(tmp_path / "im*.png"), False)
    for i in range(n):
        x = ds[i]
        assert x.dtype == torch.float32 and x.is_contiguous() and torch.equal(x, T(g["img_out"][i]))
    random.seed(123)
    dc = open_dataset(str(tmp_path / "cond.json"), True, 2)
    assert [int(os.path.basename(p)[2:-4]) for p, _ in dc.dataset] == g["cond_order"].tolist()
    for i in range(n):
        x, lab = dc[i]
        assert torch.equal(x, T(g["cond_img_out"][i])) and torch.equal(lab, T(g["cond_labels_out"][i])) and lab.dtype == torch.float32
    random.seed(321)
    dd = open_pair_dataset(str(tmp_path / "doodle.json"))
    assert [int(os.path.basename(p)[2:-4]) for p, _ in dd.dataset] == g["doodle_order"].tolist()
    for i in range(n):
        x, lab = dd[i]
        assert torch.equal(x, T(g["doodle_img_out"][i])) and torch.equal(lab, T(g["doodle_lab_out"][i]))


def test_philox_known_answers_and_normal_moments():
    """Philox4x32-10 (the generator sdm_ddpm_update draws the DDPM step's z from) against the three known-answer vectors
    published with Random123 (kat_vectors: philox4x32 10), and the Box-Muller normals' first moments."""
    def h(*a):
        return np.array(a, dtype=np.uint32)[None]
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        assert [int(v) for v in O.philox4x32_10(h(*ctr), h(*key))[0]] == list(want)
    z = O.philox_normal(0x1234_5678_9abc_def0, 977, 1 << 18)
    assert abs(z.mean()) < 1e-2 and abs(z.std() - 1) < 1e-2 and abs((z ** 4).mean() - 3) < 0.1 and abs((z ** 3).mean()) < 0.05
    assert not np.allclose(z[:64], O.philox_normal(0x1234_5678_9abc_def0, 978, 64))          # another step: another stream
    assert np.array_equal(z[:999], O.philox_normal(0x1234_5678_9abc_def0, 977, 999))          # prefix-stable in n


@pytest.mark.parametrize("case", CASES)
def test_storage_modes_of_the_oracle(case):
    """The bf16-STORAGE mode of the oracle (round 4: the pin of the HIP bf16 paths, see the block comment in
    oracle/unet_oracle.py): storage=torch.float32 is the golden-pinned arithmetic bit for bit (forward and gradients);
    storage=torch.bfloat16 differs from it by bf16 storage noise only -- a relative L2 of 2^-9 / sqrt(3) ~ 1e-3 per stored
    tensor, ~sqrt(#stages) of that at the output: between 5e-4 and 2e-2 for these nets -- and keeps the set of dead
    parameters; its training form rounds gradients where the HIP path stores them in bf16."""
    g = load(f"model_{case}.npz")
    cfg = json.loads(str(g["cfg"]))
    P = O.seeded_params(cfg, seed=3, gain=float(g["gain"]))
    x, t = T(g["x"]), T(g["t"])
    cond = T(g["cond"]) if "cond" in g else None
    tt = t if cfg["time_dim"] is not None else None
    with torch.no_grad():
        y0 = O.unet_forward(P, cfg, x, tt, cond)
        y1 = O.unet_forward(P, cfg, x, tt, cond, storage=torch.float32)
        yb = O.unet_forward(P, cfg, x, tt, cond, storage=torch.bfloat16)
    assert torch.equal(y0, y1)
    assert torch.allclose(y0, T(g["y"]), rtol=1e-4, atol=2e-6)
    r = float((yb - y0).norm() / y0.norm())
    assert 5e-4 < r < 2e-2, r
    grads = {}
    for name, st in (("f32", torch.float32), ("bf16", O.Bf16Storage(train=True))):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        loss = torch.nn.functional.mse_loss(O.unet_forward(Pg, cfg, x, tt, cond, storage=st), T(g["target"]))
        loss.backward()
        grads[name] = (float(loss), Pg)
    assert abs(grads["f32"][0] - float(g["loss"])) < 1e-5
    assert abs(grads["bf16"][0] - float(g["loss"])) < 1e-2 * max(1.0, abs(float(g["loss"])))
    errs = []
    for k in P:
        a, b = grads["bf16"][1][k].grad, grads["f32"][1][k].grad
        assert (a is None) == (b is None), k
        if b is None or (".conv_block_" in k and k.endswith("conv_layer.0.bias")):
            continue            # (conv biases ahead of a GroupNorm: a cancelling sum, ~1e-7 of its terms)
        errs.append(float((a - b).norm() / (b.norm() + 1e-30)))
    errs.sort()
    assert 1e-4 < errs[len(errs) // 2] < 2e-2 and errs[-1] < 0.15, (errs[len(errs) // 2], errs[-1])


def test_storage_model_constants_follow_the_hip_path():
    """Bf16Storage decides where attention probabilities are NOT rounded (the short-sequence no-grad core never writes
    them) from the same constants as sdm_hip/ops.py:_small_attn_ok."""
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simple-diffusion-model_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    from sdm_hip import ops
    q = O.Bf16Storage()
    assert (q.max_s, q.max_work) == (ops.SMALL_ATTN_MAX_S, ops.SMALL_ATTN_MAX_WORK) and q.keep_p_max_s == 64
    assert not q.p_is_rounded(1, 1, 256, 512) and q.p_is_rounded(4, 1, 256, 512) and q.p_is_rounded(1, 1, 1024, 512)
    assert O.Bf16Storage(train=True).p_is_rounded(1, 1, 64, 512)
