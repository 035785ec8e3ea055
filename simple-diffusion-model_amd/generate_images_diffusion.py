"""Sample images with DDPM / DDIM on MI355X — same command line and inference ``config.json`` schema as the
reference's generate_images_diffusion.py (:41-103 flags, :153-254 "models" list for ensembles/cascades).
``--device cpu`` is accepted by the parser for compatibility but refused at run time: the U-Net kernels exist
for gfx950 only."""
import argparse
import json
import os
import pathlib
import uuid
from datetime import datetime

import numpy as np
import torch

from degraders import CosineNoiseDegradation, NoiseDegradation
from diffusion_enums import DiffusionAlg, NoiseScheduler
from diffusion_sampling_algorithms import ddim_sampling, ddpm_sampling
from models.U_Net import U_Net
from utils.utils import load_checkpoint, plot_sampled_images

_FLAGS = [
    (("--device",), dict(choices=["cpu", "cuda"], type=str, default="cpu", help="Hardware device (the kernels need 'cuda').")),
    (("-c", "--config"), dict(required=True, type=pathlib.Path, help="File path to config file.")),
    (("-s", "--seed"), dict(type=int, default=None, help="Seed value for generating image (default: None).")),
    (("-n", "--num_images"), dict(type=int, default=1, help="Number of images to generate (default=1).")),
    (("-d", "--dest_path"), dict(type=pathlib.Path, help="Folder for the generated grid (default: ./plots).")),
    (("--diff_alg",), dict(default="ddpm", choices=[a.name.lower() for a in DiffusionAlg], help="Sampling algorithm.")),
    (("--ddim_step_size",), dict(type=int, default=10, help="Steps skipped per DDIM update.")),
    (("-T", "--max_T"), dict(type=int, default=1_000, help="Max T of the noise schedule (ensembles).")),
    (("--cond_img_path",), dict(type=pathlib.Path, default=None, help="Conditional image, e.g. a doodle.")),
    (("-l", "--labels"), dict(nargs="*", type=float, default=None, help="Conditional labels.")),
    # extension (not in the reference): classifier-free guidance weight; omit for the reference behaviour
    (("--cfg_scale",), dict(type=float, default=None, help="Classifier-free guidance weight w (conditional models).")),
    # extension (not in the reference): bf16 storage of activations / contraction weights in the U-Net forward
    (("--bf16",), dict(action="store_true", help="Run the U-Net in bf16 storage (fp32 accumulation, sampler in fp32).")),
]

_UNET_KEYS = dict(in_channel="in_channel", out_channel="out_channel", num_layers="num_layers",
                  num_resnet_blocks="num_resnet_block", attn_layers="attn_layers", num_heads="attn_heads",
                  dim_per_head="attn_dim_per_head", time_dim="time_dim", cond_dim="cond_dim",
                  min_channel="min_channel", max_channel="max_channel", image_recon="image_recon")


def _cond_image(path, count):
    from PIL import Image
    if not os.path.isfile(path):
        raise FileNotFoundError("Invalid path for conditional image, kindly correct and try again!")
    bgr = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)[:, :, ::-1]
    img = torch.from_numpy(((bgr - 127.5) / 127.5).copy()).permute(2, 0, 1)
    return img.unsqueeze(0).repeat(count, 1, 1, 1)


def generate_images_diffusion(raw_args=None, log=print, cond_img=None, save_locally=True):
    parser = argparse.ArgumentParser(description="Generate Images using Diffusion models.")
    for names, kw in _FLAGS:
        parser.add_argument(*names, **kw)
    args = vars(parser.parse_args(raw_args))

    if args["seed"] is not None:
        torch.manual_seed(args["seed"])
    if args["num_images"] <= 0:
        raise ValueError("Invalid image numbers, should be greater than 0!")
    if args["dest_path"] is not None and not args["dest_path"].exists():
        raise ValueError("Invalid destination path, kindly correct and ensure it exists!")
    out_dir = "./" if args["dest_path"] is None else str(args["dest_path"])
    ddim = args["diff_alg"] == DiffusionAlg.DDIM.name.lower()
    if ddim and not (0 <= args["ddim_step_size"] <= args["max_T"]):
        raise ValueError("Invalid step size for DDIM!")
    if args["device"] != "cuda":
        raise RuntimeError("This build runs the U-Net on MI355X only: pass --device cuda")
    dev = torch.device("cuda")

    if args["cond_img_path"] is not None:
        cond_img = _cond_image(args["cond_img_path"], args["num_images"])
    elif cond_img is not None:
        if not isinstance(cond_img, np.ndarray):
            raise ValueError("Unsupported conditional image.")
        cond_img = torch.from_numpy((cond_img.astype(float) - 127.5) / 127.5).float().permute(2, 0, 1)
        cond_img = cond_img.unsqueeze(0).repeat(args["num_images"], 1, 1, 1)

    with open(args["config"], "r") as f:
        details = json.load(f)
    if "models" not in details or len(details["models"]) == 0:
        raise ValueError("Invalid/no model details in json, kindly correct and try again!")
    folder = os.path.split(args["config"])[0]

    x_t, shape = None, None
    for md in details["models"]:                 # ensemble / cascade: x_t is carried from model to model
        if x_t is None:
            shape = (args["num_images"], md["img_C"], md["img_H"], md["img_W"])
            x_t = torch.randn(shape, device=dev)
        labels = None
        if md["cond_dim"] is not None:
            if args["labels"] is None or len(args["labels"]) != md["cond_dim"]:
                raise ValueError("Invalid / No conditional labels passed!")
            labels = torch.tensor(args["labels"]).float().to(dev)          # shape (cond_dim,), broadcast over the batch
        if md["noise_scheduler"] == NoiseScheduler.LINEAR.name:
            schedule = NoiseDegradation(md["beta_1"], md["beta_T"], args["max_T"], dev)
        elif md["noise_scheduler"] == NoiseScheduler.COSINE.name:
            schedule = CosineNoiseDegradation(args["max_T"])
        else:
            raise ValueError("Invalid noise scheduler type.")
        net = U_Net(**{k: md[v] for k, v in _UNET_KEYS.items()}).to(dev)
        model_path = os.path.join(folder, md["model_name"])
        if not os.path.isfile(model_path):
            raise FileNotFoundError("Invalid path for model in json file, kindly correct and try again!")
        ok, ckpt = load_checkpoint(model_path)
        if not ok:
            raise Exception("Failed to load model!")
        net.load_state_dict(ckpt["model"])
        if args["bf16"]:
            net.set_inference_dtype(torch.bfloat16)
        common = dict(diffusion_net=net, noise_degradation=schedule, x_t=x_t, min_noise=md["min_noise"],
                      max_noise=md["max_noise"], cond_img=cond_img, labels_tensor=labels, device=dev, log=log,
                      cfg_scale=args["cfg_scale"] if labels is not None else None)
        x_t = ddim_sampling(ddim_step_size=args["ddim_step_size"], **common) if ddim else ddpm_sampling(**common)

    if not save_locally:
        return x_t
    name = f"{datetime.now().strftime('%d-%m-%Y %H:%M:%S')}_({shape[2]},{shape[3]})_{uuid.uuid4().hex}"
    plot_sampled_images(sampled_imgs=x_t, file_name=name, dest_path=out_dir, log=log)
    return None


if __name__ == "__main__":
    generate_images_diffusion()
