"""CPU-side checks: the C-ABI library loads and exports every symbol include/sdm_hip.h declares, struct layouts
agree, the drop-in Python surface has the reference's names / exceptions / state_dict keys, and the product path
refuses to run without the GPU (no silent fallback)."""
import json
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from sdm_hip import lib
    hdr = open(os.path.join(ROOT, "include", "sdm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|size_t|const char\*|unsigned long long)\s+(sdm_\w+)\s*\(", hdr, re.M))
    assert len(declared) >= 30
    missing = [n for n in sorted(declared) if not hasattr(lib._lib, n)]
    assert not missing, missing
    unbound = sorted(declared - set(lib.EXPORTS))
    assert not unbound, f"declared in the header but not bound in sdm_hip/lib.py: {unbound}"
    assert lib.version() >= 200
    # struct layout agreement is asserted at import; re-check one explicitly
    import ctypes as C
    assert lib._lib.sdm_struct_size(0) == C.sizeof(lib.IgemmDesc)


def test_launch_functions_validate_arguments_without_a_gpu():
    """Argument validation happens before any HIP call, so error paths are testable on CPU."""
    import ctypes as C
    from sdm_hip import lib
    d = lib.IgemmDesc()
    with pytest.raises(lib.SdmError, match="igemm"):
        lib.call("sdm_igemm", C.byref(d), None)
    with pytest.raises(lib.SdmError, match="col_softmax"):
        lib.call("sdm_col_softmax", None, 0, 0, 0, None)
    # bf16 entry points: same descriptor, stricter alignment rules, checked before any launch
    with pytest.raises(lib.SdmError, match="igemm_bf16"):
        lib.call("sdm_igemm_bf16", C.byref(d), 0, None)
    d.M, d.N, d.K, d.batch, d.bdiv, d.Ht, d.Wt, d.nphase = 64, 64, 36, 1, 1, 64, 1, 1      # K not a multiple of 8
    d.ntaps[0] = 1
    d.a.p0 = d.b = d.out = 0x1000
    d.a.csplit = 1 << 30
    with pytest.raises(lib.SdmError, match="multiples of 8"):
        lib.call("sdm_igemm_bf16", C.byref(d), 0, None)
    with pytest.raises(lib.SdmError, match="affine_residual_bf16"):
        lib.call("sdm_affine_residual_bf16", None, 0, None, None, 0, None, 0, None, 0, 0, None, 0, 1, 1, 8, None)
    with pytest.raises(lib.SdmError, match="transpose_bf16"):
        lib.call("sdm_transpose_bf16", None, 0, 0, 0, 1, None, 0, 0, 1, 1, 1, None)


def test_unet_surface_matches_reference():
    from models.U_Net import U_Net
    from oracle import unet_oracle as O
    with pytest.raises(TypeError):
        U_Net(num_layers="4")
    with pytest.raises(TypeError):
        U_Net(attn_layers=(1, 2))
    with pytest.raises(ValueError):
        U_Net(num_layers=0, attn_layers=[])
    with pytest.raises(ValueError):
        U_Net(num_layers=2, attn_layers=[2])
    with pytest.raises(ValueError):
        U_Net(num_layers=2, attn_layers=[0.5])
    cfg = dict(num_resnet_blocks=2, in_channel=6, out_channel=3, time_dim=32, cond_dim=4, num_layers=3, attn_layers=[0, 2],
               num_heads=2, dim_per_head=8, min_channel=32, max_channel=64, image_recon=True)
    net = U_Net(**cfg)
    shapes = O.param_shapes(O.full_cfg(cfg))
    sd = net.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    dead = set(O.dead_param_names(O.full_cfg(cfg)))
    assert {n for n, _ in net.named_parameters()} - {n for n, _ in net.live_parameters()} == dead
    # tap-major storage behind PyTorch-shaped views
    w = net.in_layer[1].conv_layer.conv.weight
    assert w.shape == (32, 32, 3, 3) and w.permute(2, 3, 0, 1).is_contiguous()
    wt = net.up_layers[0].out_layer.conv_layer.conv.weight
    assert wt.shape == (128, 64, 4, 4) and wt.permute(2, 3, 1, 0).is_contiguous()
    # lenient loader skips unknown names and mismatched shapes, strict loader accepts reference-shaped tensors
    ref_like = {k: torch.randn(v) for k, v in shapes.items()}
    net.load_state_dict(ref_like, strict=True)
    assert torch.equal(net.state_dict()["in_layer.1.conv_layer.0.weight"], ref_like["in_layer.1.conv_layer.0.weight"])
    assert w.permute(2, 3, 0, 1).is_contiguous()            # layout survives load_state_dict
    bad = dict(ref_like, bogus=torch.zeros(1))
    bad["out_layers.1.conv_layer.0.bias"] = torch.zeros(7)
    net.custom_load_state_dict(bad)
    # flat gradient buffer covers the live parameters once, in completion order
    flat, grads, ranges = net.alloc_grads()
    assert flat.numel() == sum(p.numel() for _, p in net.live_parameters())
    assert list(ranges)[0] == "out_layers" and list(ranges)[-1] == "cond_emb"
    assert all(grads[p].stride() == p.stride() for _, p in net.live_parameters())


def test_no_cpu_fallback():
    from models.U_Net import U_Net
    net = U_Net(num_resnet_blocks=1, num_layers=1, attn_layers=[], time_dim=32, min_channel=32, max_channel=32)
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.randn(1, 3, 8, 8), torch.tensor([5]))
    import degraders
    with pytest.raises(RuntimeError, match="MI355X"):
        degraders.CosineNoiseDegradation(1000)(torch.randn(1, 3, 8, 8), torch.tensor([5]))
    import diffusion_sampling_algorithms as S
    with pytest.raises(RuntimeError, match="MI355X"):
        S.ddpm_sampling(net, degraders.CosineNoiseDegradation(1000), torch.randn(1, 3, 8, 8))


def test_generator_cli_argument_contract(tmp_path):
    import generate_images_diffusion as G
    with pytest.raises(SystemExit):
        G.generate_images_diffusion([])                      # -c/--config is required
    cfg = tmp_path / "config.json"
    cfg.write_text(json.dumps({"models": []}))
    with pytest.raises(ValueError, match="image numbers"):
        G.generate_images_diffusion(["-c", str(cfg), "-n", "0"])
    with pytest.raises(ValueError, match="DDIM"):
        G.generate_images_diffusion(["-c", str(cfg), "--diff_alg", "ddim", "--ddim_step_size", "2000"])
    with pytest.raises(RuntimeError, match="cuda"):
        G.generate_images_diffusion(["-c", str(cfg), "--device", "cpu"])


def test_ddim_step_list_and_schedule_tables():
    import degraders as D
    lin = D.NoiseDegradation(5e-3, 9e-3, 1000)
    b, a, ab = lin.get_timestep_params(torch.tensor([0, 1, 1000]))
    assert float(b[0]) == pytest.approx(5e-3) and float(b[2]) == pytest.approx(9e-3)
    assert float(ab[1]) == pytest.approx((1 - 5e-3) * float(a[1]))
    cos = D.CosineNoiseDegradation(1000)
    b, a, ab = cos.get_timestep_params(torch.tensor([1, 500, 1000]))
    assert float(ab[0]) == pytest.approx(0.99995875, abs=1e-6) and float(b[2]) == pytest.approx(0.999)
    assert cos.host_params(500)[2] == pytest.approx(0.49384364, abs=1e-6)


def test_grad_blocks_cover_live_parameters_once_and_keep_y_scale_last():
    """The flat gradient buffer / data-parallel buckets follow U_Net.grad_blocks(): every live parameter exactly once,
    and the AdaGN y_scale projections -- still read by the batched embedding-gradient launch at the end of the backward
    -- only in the final block, so that a per-bucket optimizer update can never run ahead of a gradient that needs them."""
    from models.U_Net import U_Net
    net = U_Net(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=32, cond_dim=3, num_layers=2, attn_layers=[1],
                num_heads=2, dim_per_head=16, min_channel=32, max_channel=64)
    blocks = net.grad_blocks()
    listed = [id(p) for _, ps in blocks for p in ps]
    live = [id(p) for _, p in net.live_parameters()]
    assert sorted(listed) == sorted(live) and len(set(listed)) == len(listed)
    names = {id(p): n for n, p in net.named_parameters()}
    assert blocks[-1][0] == "cond_emb"
    for bname, ps in blocks[:-1]:
        assert not any(".adagn.y_scale." in names[id(p)] for p in ps), bname
    assert sum(".adagn.y_scale." in names[id(p)] for p in blocks[-1][1]) == 2 * len(net._adagns) > 0
    # a model without embedding has no late parameters and no cond_emb block
    plain = U_Net(num_resnet_blocks=1, in_channel=3, out_channel=3, time_dim=None, cond_dim=None, num_layers=2,
                  attn_layers=[], num_heads=1, dim_per_head=None, min_channel=32, max_channel=64)
    pb = plain.grad_blocks()
    assert sorted(id(p) for _, ps in pb for p in ps) == sorted(id(p) for _, p in plain.live_parameters())


def test_linear_small_rejects_an_addend_that_does_not_cover_the_batch():
    """a 1-row time embedding added to n label rows used to be read as n rows.  The addend now travels
    with its row count and pitch, and anything but M rows / one broadcast row is refused before any launch."""
    from sdm_hip import lib
    p = 0x1000
    with pytest.raises(lib.SdmError, match="addend has 1 rows"):
        lib.call("sdm_linear_small", p, 32, p, p, p, 32, 1, p, 32, None, 4, 32, 32, 0, None)     # 1 row but pitch != 0
    with pytest.raises(lib.SdmError, match="addend has 2 rows"):
        lib.call("sdm_linear_small", p, 32, p, p, p, 32, 2, p, 32, None, 4, 32, 32, 0, None)


def test_conditional_dataset_reads_the_reference_tinydb_schema(tmp_path):
    """custom_dataset/conditional_img_dataset.py:19-36: table "Data" (filename + one key per label) and table "Labels"
    ({"labels": [...]}); same exceptions when a table is empty."""
    import numpy as np
    from PIL import Image
    from custom_dataset.img_dataset import ConditionalImgDataset, open_dataset
    rgb = np.zeros((4, 6, 3), dtype=np.uint8)
    rgb[..., 0], rgb[..., 2] = 255, 0                     # pure red in RGB
    rgb[..., 1] = 127
    for name in ("a.png", "b.png"):
        Image.fromarray(rgb).save(tmp_path / name)
    db = {"Data": {"1": {"filename": str(tmp_path / "a.png"), "smiling": 1, "glasses": 0.0},
                   "2": {"filename": "b.png", "smiling": 0, "glasses": 1}},
          "Labels": {"1": {"labels": ["smiling", "glasses"]}}}
    path = tmp_path / "db.json"
    path.write_text(json.dumps(db))
    ds = open_dataset(str(path), True, 2)
    assert isinstance(ds, ConditionalImgDataset) and len(ds) == 2 and ds.get_labels() == ["smiling", "glasses"]
    seen = {}
    for i in range(2):
        img, lab = ds[i]
        assert img.shape == (3, 4, 6) and img.dtype == torch.float32 and lab.dtype == torch.float32
        assert float(img[0].max()) == -1.0 and float(img[2].min()) == 1.0          # BGR order, scaled to [-1, 1]
        seen[tuple(lab.tolist())] = True
    assert seen == {(1.0, 0.0): True, (0.0, 1.0): True}
    (tmp_path / "empty.json").write_text(json.dumps({"Data": {}, "Labels": db["Labels"]}))
    with pytest.raises(Exception, match="No data found in Data table"):
        ConditionalImgDataset(str(tmp_path / "empty.json"))
    (tmp_path / "nolabels.json").write_text(json.dumps({"Data": db["Data"]}))
    with pytest.raises(Exception, match="No data found in Labels table"):
        ConditionalImgDataset(str(tmp_path / "nolabels.json"))


def test_normalisation_table_matches_the_reference_arithmetic():
    """img_dataset.py:29-32: (img.astype(float) - 127.5) / 127.5 in float64, then .float(): the 256-entry table the device
    input pipeline gathers through holds exactly those values."""
    import numpy as np
    from custom_dataset.device_cache import norm_table
    v = np.arange(256, dtype=np.uint8)
    ref = torch.from_numpy((v.astype(float) - 127.5) / 127.5).float()
    assert torch.equal(norm_table(), ref)
    assert float(ref[0]) == -1.0 and float(ref[255]) == 1.0


def _bench(args, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None) if "WORLD_SIZE" not in env_extra else None
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus N` is what the driver runs: the process must start N ranks ITSELF (the flag used
    to be parsed and ignored).  Dry run (no GPU): the ranks rendezvous over gloo and count themselves."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"SDM_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["ranks_counted"] == 2 and out["gpus_arg"] == 2
    # an external launcher that disagrees with --gpus is an error, not a silently different job
    r = _bench(["--gpus", "2"], {"SDM_BENCH_DRYRUN": "1", "WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
    # a failing rank fails the launcher (here: no GPU for the real run)
    if not torch.cuda.is_available():
        r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-configs", "--no-cpu-baseline"], {})
        assert r.returncode != 0


def test_dataset_record_order_is_rank_consistent_under_data_parallelism(tmp_path, monkeypatch):
    """the conditional / doodle datasets shuffle their records at construction (as the reference does).  Under
    WORLD_SIZE > 1 every rank builds its own dataset, and the samplers hand out index shards: the shards are disjoint and
    cover the dataset only if the record order is the same on every rank, whatever each process's `random` state is."""
    import random
    from custom_dataset.img_dataset import ConditionalImgDataset, DoodleImgDataset
    n = 24
    db = {"Data": {str(i + 1): {"filename": f"im{i:02d}.png", "lab": f"lab{i:02d}.png", "a": float(i)} for i in range(n)},
          "Labels": {"1": {"labels": ["lab", "a"]}}}
    (tmp_path / "db.json").write_text(json.dumps(db))
    dbc = {"Data": db["Data"], "Labels": {"1": {"labels": ["a"]}}}
    (tmp_path / "dbc.json").write_text(json.dumps(dbc))
    monkeypatch.setenv("WORLD_SIZE", "2")
    for cls, path in ((ConditionalImgDataset, "dbc.json"), (DoodleImgDataset, "db.json")):
        per_rank = []
        for rank in range(2):
            random.seed(1000 + rank)                       # ranks are separate processes with unrelated `random` states
            ds = cls(str(tmp_path / path))
            sampler = torch.utils.data.distributed.DistributedSampler(ds, 2, rank, shuffle=True)
            sampler.set_epoch(5)
            per_rank.append({ds.dataset[i][0] for i in sampler})
        assert len(per_rank[0]) == len(per_rank[1]) == n // 2
        assert not (per_rank[0] & per_rank[1]) and len(per_rank[0] | per_rank[1]) == n
    # one process: the reference's plain random.shuffle (seedable through `random`, and really shuffled)
    monkeypatch.delenv("WORLD_SIZE")
    random.seed(7)
    a = [r[0] for r in ConditionalImgDataset(str(tmp_path / "dbc.json")).dataset]
    random.seed(8)
    b = [r[0] for r in ConditionalImgDataset(str(tmp_path / "dbc.json")).dataset]
    assert sorted(a) == sorted(b) and a != b


def test_argument_validation_under_asan_ubsan(tmp_path):
    """SURVEY §5 "race detection / sanitizers": the entry points' argument-validation and bookkeeping paths (no GPU call is
    reached) under a host AddressSanitizer + UndefinedBehaviorSanitizer build of the light translation units
    (`make -C csrc asan`; device code is not instrumented).  Runs in a child process with the sanitizer runtime preloaded; any
    report fails the child (halt_on_error)."""
    import subprocess
    import glob
    csrc = os.path.join(ROOT, "simple-diffusion-model_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    so = os.path.join(ROOT, "simple-diffusion-model_amd", "sdm_hip", "libsdm_hip_asan.so")
    rt = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no ASan runtime in this image")
    child = r"""
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
L.sdm_last_error.restype = C.c_char_p
p = C.c_void_p
def call(name, *a):
    rc = getattr(L, name)(*a)
    return rc, L.sdm_last_error().decode()
bad = 0x1000
checks = [
  ("sdm_col_softmax", (p(None), 0, 0, C.c_int64(0), p(None)), "col_softmax"),
  ("sdm_linear_small", (p(bad), C.c_int64(32), p(bad), p(bad), p(bad), C.c_int64(32), 2, p(bad), C.c_int64(32), p(None), 4, 32, 32, 0, p(None)), "addend has 2 rows"),
  ("sdm_step_begin", (p(None), p(None), p(None), 3, 5, p(None), p(None), p(None), p(None), p(None), 0, 0, p(None)), "step_begin"),
  ("sdm_step_begin", (p(bad), p(bad), p(bad), 3, 5, p(bad), p(bad), p(bad), p(None), p(bad), 1, 6, p(None)), "width"),
  ("sdm_ddpm_update", (p(None), p(None), p(None), p(None), C.c_size_t(0), C.c_ulonglong(1), p(None), p(None), 0, p(None)), "ddpm_update"),
  ("sdm_ddim_update", (p(bad), p(bad), p(bad), p(None), C.c_size_t(4), p(bad), 0, p(None)), "cursor without nsteps"),
  ("sdm_attn_scores_small", (p(bad), C.c_int64(48), 1, 300, 1, 16, C.c_float(1.0), p(bad), C.c_int64(300), 0, p(None)), "at most 256"),
  ("sdm_attn_scores_small", (p(bad), C.c_int64(48), 1, 16, 1, 12, C.c_float(1.0), p(bad), C.c_int64(16), 0, p(None)), "multiple of 16"),
  ("sdm_attn_small_nograd", (p(bad), C.c_int64(48), 1, 16, 1, 16, C.c_float(1.0), p(bad), C.c_int64(16), 0, p(None), C.c_size_t(0), p(None)), "attn_small_nograd"),
  ("sdm_gn_apply_coop", (p(bad), C.c_int64(128), 0, C.c_float(1.0), p(None), 0, 1, 4096, 128, 32, C.c_float(1e-5), p(bad), p(bad), p(None), 0, p(None), C.c_int64(0), p(None), C.c_int64(0), 0, p(bad), C.c_int64(128), 0, p(None), C.c_size_t(0), p(None)), "state buffer"),
  ("sdm_plan_end", (), "no plan is being recorded"),
  ("sdm_plan_run", (7, 1, p(None)), "bad handle"),
  ("sdm_prof_begin", (0,), "capacity"),
]
for name, args, frag in checks:
    rc, err = call(name, *args)
    assert rc < 0 and frag in err, (name, rc, err)
assert L.sdm_plan_begin() == 0 and L.sdm_plan_begin() < 0            # second recording refused
assert L.sdm_plan_end() == 0 and "no launches" in L.sdm_last_error().decode()
L.sdm_struct_size.restype = C.c_size_t
assert L.sdm_struct_size(0) > 400 and L.sdm_struct_size(99) == 0
L.sdm_gn_apply_coop_state_bytes.restype = C.c_size_t
assert L.sdm_gn_apply_coop_state_bytes() > 0 and L.sdm_version() >= 300
print("asan-ok")
"""
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", child, so], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
