"""Shared training command-line body (see train_diffusion.py / train_SR_diffusion.py for the user-facing docs)."""
import argparse
import csv
import json
import logging
import os
import pathlib

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from custom_dataset.img_dataset import open_dataset, open_pair_dataset
from degraders import CosineNoiseDegradation, NoiseDegradation
from diffusion_enums import DiffusionAlg, NoiseScheduler
from diffusion_sampling_algorithms import cold_diffusion_sampling, ddim_sampling, ddpm_sampling
from models.U_Net import U_Net
from sdm_hip import ops
from sdm_hip.trainer import DiffusionTrainer
from utils.utils import load_checkpoint, plot_sampled_images, save_model

SR_KEYS = ["lr_dim", "sr_dim", "cond_t"]          # train_SR_diffusion.py:60-62
REQUIRED_KEYS = ["checkpoint_steps", "lr_steps", "max_epoch", "plot_img_count", "use_conditional", "flip_imgs", "dataset_path",
                 "out_dir", "model_checkpoint", "config_checkpoint", "load_diffusion_optim", "diffusion_lr", "batch_size",
                 "noise_scheduler", "min_noise_step", "max_noise_step", "max_actual_noise_step", "skip_step",
                 "in_channel", "out_channel", "num_layers", "num_resnet_block", "attn_layers", "attn_heads",
                 "attn_dim_per_head", "time_dim", "cond_dim", "min_channel", "max_channel", "img_recon"]


def _schedule(cfg, beta_1, beta_T, device):
    if cfg["noise_scheduler"] == NoiseScheduler.LINEAR.name:
        return NoiseDegradation(beta_1, beta_T, cfg["max_noise_step"], device)
    return CosineNoiseDegradation(cfg["max_noise_step"])


def _save(out_dir, net, opt, steps, starting_epoch, linear, beta_1, beta_T):
    state = {"starting_epoch": starting_epoch, "global_steps": steps}      # the reference stores the INITIAL epoch (:375-377)
    if linear:
        state.update(beta_1=beta_1, beta_T=beta_T)
    save_model(model_net=state, file_name="config", dest_path=out_dir, checkpoint=True, steps=steps)
    save_model(model_net={"model": net.state_dict(), "optimizer": opt.state_dict()}, file_name="diffusion",
               dest_path=out_dir, checkpoint=True, steps=steps)


PROJECT = {"diffusion": "Diffusion", "sr": "Diffusion", "cold": "Noise-Cold-Diffusion", "doodle": "Doodle-Diffusion"}   # log file names


def run_training(sr=False, mode=None):
    """Shared body of the four trainers: train_diffusion.py ("diffusion"), train_SR_diffusion.py ("sr"),
    train_noise_cold_diffusion.py ("cold": x0 target, cold sampler for the plots) and train_doodle_diffusion.py ("doodle":
    image-conditioned, the label image is concatenated to x_t on the channel axis)."""
    mode = mode or ("sr" if sr else "diffusion")
    sr = mode == "sr"
    cold, doodle = mode == "cold", mode == "doodle"
    project_name = PROJECT[mode]
    parser = argparse.ArgumentParser(description="Train Diffusion models.")
    parser.add_argument("-c", "--config-path", help="File path to load json config file.", required=True, type=pathlib.Path)
    parser.add_argument("--device", help="Which hardware device will model run on (default='cpu')?", choices=["cpu", "cuda"],
                        type=str, default="cpu")
    args = vars(parser.parse_args())
    if args["device"] != "cuda":
        raise RuntimeError("This build trains on MI355X only: pass --device cuda")

    with open(args["config_path"], "r") as f:
        cfg = json.loads(f.read())
    required = [k for k in REQUIRED_KEYS if not (doodle and k in ("use_conditional", "flip_imgs"))]
    for k in required + (SR_KEYS if sr else []):        # the reference indexes every key without defaults -> KeyError
        cfg[k]
    if doodle:                                           # train_doodle_diffusion.py has neither switch
        cfg.setdefault("use_conditional", False)
        cfg.setdefault("flip_imgs", False)
    if cfg["dataset_path"] is None:
        raise ValueError("No dataset_path entered.")
    if cfg["noise_scheduler"] not in (NoiseScheduler.LINEAR.name, NoiseScheduler.COSINE.name):
        raise ValueError("Invalid noise scheduler type.")
    if not sr and not cold and cfg["diffusion_alg"] not in (DiffusionAlg.DDIM.name, DiffusionAlg.DDPM.name):
        raise ValueError("Invalid diffusion algorithm type.")
    linear = cfg["noise_scheduler"] == NoiseScheduler.LINEAR.name
    beta_1, beta_T = (cfg["beta1"], cfg["betaT"]) if linear else (None, None)
    lo, hi, hi_act, skip = cfg["min_noise_step"], cfg["max_noise_step"], cfg["max_actual_noise_step"], cfg["skip_step"]
    if hi_act < lo or hi < lo or skip > hi_act or skip < 0 or lo < 0:
        raise ValueError("Invalid step values entered!")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())   # (rehearsals: ranks may share a GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SDM_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    chief = rank == 0

    out_dir = cfg["out_dir"]
    os.makedirs(out_dir, exist_ok=True)
    handlers = [logging.StreamHandler()] + ([logging.FileHandler(os.path.join(out_dir, f"{project_name}.log"))] if chief else [])
    logging.basicConfig(format="%(asctime)s %(message)s", handlers=handlers, level=logging.DEBUG if chief else logging.WARNING)
    if cfg.get("seed") is not None:
        torch.manual_seed(cfg["seed"])

    dataset = open_pair_dataset(cfg["dataset_path"]) if doodle else \
        open_dataset(cfg["dataset_path"], cfg["use_conditional"], cfg["cond_dim"])
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True) if world > 1 else None
    loader = torch.utils.data.DataLoader(dataset, batch_size=cfg["batch_size"], num_workers=4, shuffle=sampler is None,
                                         sampler=sampler, drop_last=world > 1, pin_memory=True)
    plot_batch = next(iter(torch.utils.data.DataLoader(dataset, batch_size=cfg["plot_img_count"], num_workers=0, shuffle=False)))
    plot_imgs, plot_labels = (plot_batch if (cfg["use_conditional"] or doodle) else (plot_batch, None))
    if doodle and chief:                                  # train_doodle_diffusion.py:146-149
        plot_sampled_images(sampled_imgs=plot_labels, file_name="label_plot", dest_path=out_dir)
    if cfg["use_conditional"] and chief:
        with open(os.path.join(out_dir, "labels.txt"), "a") as f:
            csv.writer(f).writerows([dataset.get_labels()] + plot_labels.cpu().tolist())

    net = U_Net(in_channel=cfg["in_channel"], out_channel=cfg["out_channel"], num_layers=cfg["num_layers"],
                num_resnet_blocks=cfg["num_resnet_block"], attn_layers=cfg["attn_layers"], num_heads=cfg["attn_heads"],
                dim_per_head=cfg["attn_dim_per_head"], time_dim=cfg["time_dim"], cond_dim=cfg["cond_dim"],
                min_channel=cfg["min_channel"], max_channel=cfg["max_channel"], image_recon=cfg["img_recon"])
    ckpt = None
    if cfg["model_checkpoint"] is not None:
        ok, ckpt = load_checkpoint(cfg["model_checkpoint"])
        if not ok:
            raise Exception("An error occured while loading model checkpoint!")
        net.custom_load_state_dict(ckpt["model"])
    net = net.to(dev)

    starting_epoch, global_steps = 0, 0
    if cfg["config_checkpoint"] is not None:
        ok, cc = load_checkpoint(cfg["config_checkpoint"])
        if not ok:
            raise Exception("An error occured while loading config checkpoint!")
        if linear:
            beta_1, beta_T = cc["beta_1"], cc["beta_T"]
        starting_epoch, global_steps = cc["starting_epoch"], cc["global_steps"]

    schedule = _schedule(cfg, beta_1, beta_T, dev)
    trainer = DiffusionTrainer(net, schedule, lr=cfg["diffusion_lr"], betas=(0.5, 0.999), min_noise_step=lo,
                               max_actual_noise_step=hi_act, seed=cfg.get("seed"),
                               sr=(cfg["lr_dim"], cfg["sr_dim"], cfg["cond_t"]) if sr else None,
                               objective="x0" if cold else "eps",
                               train_dtype=torch.bfloat16 if cfg.get("train_dtype") == "bf16" else None)
    cached = False
    if cfg.get("device_cache") and not doodle and not str(cfg["dataset_path"]).startswith("synthetic"):
        # extension: the decoded dataset lives in HBM as uint8; batches are gathered / normalised / flipped by one kernel
        from custom_dataset.device_cache import CachedLoader, DeviceImageCache
        cache = DeviceImageCache(dataset, dev, log=logging.info)
        loader = CachedLoader(cache, cfg["batch_size"], rank, world, flip=cfg["flip_imgs"], generator=trainer.gen,
                              conditional=cfg["use_conditional"])
        sampler, cached = loader, True          # set_epoch() every epoch at any world size: fresh permutation per epoch
        logging.info(f"device cache: {len(cache):,} images, {cache.images.numel() / 2**20:.1f} MiB of HBM")
    opt = trainer.opt
    if ckpt is not None and cfg["load_diffusion_optim"]:
        opt.load_state_dict(ckpt["optimizer"])

    info = logging.info
    info("#" * 100)
    info(f"Train Parameters: max epoch {cfg['max_epoch']:,} | dataset {cfg['dataset_path']} | out {out_dir} | "
         f"checkpoint steps {cfg['checkpoint_steps']} | batch {cfg['batch_size']:,} x {world} GPU(s) | "
         f"LR {opt.param_groups[0]['lr']:.5f} | conditional {cfg['use_conditional']} | flip {cfg['flip_imgs']}")
    info(f"Model Parameters: in {cfg['in_channel']} out {cfg['out_channel']} layers {cfg['num_layers']} resnet "
         f"{cfg['num_resnet_block']} attn {cfg['attn_layers']} heads {cfg['attn_heads']} dim/head {cfg['attn_dim_per_head']} "
         f"time {cfg['time_dim']} cond {cfg['cond_dim']} channels {cfg['min_channel']}..{cfg['max_channel']} "
         f"recon {cfg['img_recon']}")
    info(f"Diffusion Parameters: {cfg['noise_scheduler']} {'' if not linear else (beta_1, beta_T)} steps {lo}..{hi} "
         f"(trained up to {hi_act})")
    info("#" * 100)

    for epoch in range(starting_epoch, cfg["max_epoch"]):
        if sampler is not None:
            sampler.set_epoch(epoch)
        total, count = 0.0, 0
        for index, data in enumerate(loader):
            count += 1
            cond_img = None
            if doodle:
                x0, labels, cond_img = data[0].to(dev, non_blocking=True), None, data[1].to(dev, non_blocking=True).float()
            elif cfg["use_conditional"]:
                x0, labels = data[0].to(dev, non_blocking=True), data[1].to(dev, non_blocking=True).float()
            else:
                x0, labels = data.to(dev, non_blocking=True), None
            x0 = x0.float()
            if cfg["flip_imgs"] and not cached:    # per-image horizontal flip with p = 0.5 (:312-314)
                flip = torch.rand(x0.shape[0], device=dev, generator=trainer.gen) < 0.5     # per-rank stream
                x0 = torch.where(flip[:, None, None, None], x0.flip(-1), x0)
            loss = trainer.step(x0, labels, cond_img=cond_img)
            if world > 1:                          # global mean loss; a NaN on any rank reaches every rank, so all ranks
                dist.all_reduce(loss)              # raise together instead of leaving peers blocked in the next all-reduce
                loss = loss / world
            loss_v = float(loss)                   # the step's only host sync (the reference has two, :354,:366)
            if loss_v != loss_v:                   # (deviation: the fused Adam of this step has already been enqueued; the
                raise Exception("NaN encountered during training")   # reference raises before optimizer.step, :354-358)
            total += loss_v

            if global_steps % cfg["lr_steps"] == 0 and global_steps > 0:
                for group in opt.param_groups:
                    group["lr"] = group["lr"] * 0.5

            if global_steps % cfg["checkpoint_steps"] == 0 and global_steps >= 0 and chief:
                _save(out_dir, net, opt, global_steps, starting_epoch, linear, beta_1, beta_T)
                C, H, W = x0.shape[1:]
                n_plot = cfg["plot_img_count"]
                if sr:                             # train_SR_diffusion.py:421-458
                    noise = torch.randn((n_plot, C, H, W), device=dev)
                    low = ops.area_resize(ops.area_resize(plot_imgs.to(dev).float(), (cfg["lr_dim"],) * 2), (cfg["sr_dim"],) * 2)
                    cond = schedule(low, torch.tensor([cfg["cond_t"]], device=dev), noise)
                    lab = plot_labels.to(dev).float() if plot_labels is not None else None
                    delta = cold_diffusion_sampling(diffusion_net=net, noise_degradation=schedule, x_t=1 * noise, noise=noise,
                                                    min_noise=lo, max_noise=hi_act, cond_img=cond, labels_tensor=lab,
                                                    skip_step_size=skip, device=dev, log=print)
                    plot_sampled_images(sampled_imgs=delta + low, file_name=f"diffusion_plot_{global_steps}", dest_path=out_dir)
                elif cold:                         # train_noise_cold_diffusion.py:391-419
                    noise = torch.randn((n_plot, C, H, W), device=dev)
                    x_plot = schedule(plot_imgs.to(dev).float(), torch.tensor([hi_act], device=dev), noise) if hi_act < hi \
                        else 1 * noise
                    lab = plot_labels.to(dev).float() if plot_labels is not None else None
                    imgs = cold_diffusion_sampling(diffusion_net=net, noise_degradation=schedule, x_t=x_plot, noise=noise,
                                                   min_noise=lo, max_noise=hi_act, cond_img=None, labels_tensor=lab,
                                                   skip_step_size=skip, device=dev, log=print)
                    plot_sampled_images(sampled_imgs=imgs, file_name=f"diffusion_plot_{global_steps}", dest_path=out_dir)
                elif hi_act < hi:
                    x_plot = schedule(plot_imgs.to(dev).float(), torch.tensor([hi_act], device=dev),
                                      torch.randn((n_plot, C, H, W), device=dev))
                else:
                    x_plot = torch.randn((n_plot, C, H, W), device=dev)
                if not sr and not cold:
                    lab = plot_labels.to(dev).float() if plot_labels is not None else None
                    common = dict(diffusion_net=net, noise_degradation=schedule, x_t=x_plot, min_noise=lo, max_noise=hi_act,
                                  cond_img=lab if doodle else None, labels_tensor=None if doodle else lab, device=dev,
                                  log=print)
                    if cfg["diffusion_alg"] == DiffusionAlg.DDPM.name:
                        imgs = ddpm_sampling(**common)
                    else:
                        imgs = ddim_sampling(ddim_step_size=skip, **common)
                    plot_sampled_images(sampled_imgs=imgs, file_name=f"diffusion_plot_{global_steps}", dest_path=out_dir)

            info("Cum. Steps: {:,} | Steps: {:,} / {:,} | Diffusion: {:.5f} | LR: {:.9f}".format(
                global_steps + 1, index + 1, len(loader), total / count, opt.param_groups[0]["lr"]))
            global_steps += 1

        if chief:
            _save(out_dir, net, opt, global_steps, starting_epoch, linear, beta_1, beta_T)
        info("Epoch: {:,} | Diffusion: {:.5f} | LR: {:.9f}".format(epoch, total / max(count, 1), opt.param_groups[0]["lr"]))

    if world > 1:
        dist.destroy_process_group()


