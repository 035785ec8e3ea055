"""Super-resolution sampling (cascaded up-sampler) on MI355X -- same command line and ``config.json`` schema as
the reference's generate_sr_images_diffusion.py (:41-99 flags, :141-238 model loop): the low-resolution image is
area-interpolated to the target size, noised at the fixed ``cond_t`` and fed as a conditional image to the
x0-predicting U-Net through the cold-diffusion sampler; the result is low-res + predicted residual."""
import argparse
import json
import os
import pathlib
import uuid
from datetime import datetime

import numpy as np
import torch

from degraders import CosineNoiseDegradation, NoiseDegradation
from diffusion_enums import NoiseScheduler
from diffusion_sampling_algorithms import cold_diffusion_sampling
from generate_images_diffusion import _UNET_KEYS
from models.U_Net import U_Net
from sdm_hip import ops
from utils.utils import load_checkpoint, plot_sampled_images

_FLAGS = [
    (("--device",), dict(choices=["cpu", "cuda"], type=str, default="cpu", help="Hardware device (the kernels need 'cuda').")),
    (("-c", "--config"), dict(required=True, type=pathlib.Path, help="File path to load config file.")),
    (("-s", "--seed"), dict(type=int, default=None, help="Seed value for generating image (default: None).")),
    (("-d", "--dest_path"), dict(type=pathlib.Path, help="Folder for the generated image (default: out folder).")),
    (("--cold_step_size",), dict(type=int, default=10, help="Number of steps to skip when using cold diffusion.")),
    (("-l", "--labels"), dict(nargs="*", type=float, default=None, help="Labels Conditional e.g One-hot encoding.")),
    (("--lr_img_path",), dict(type=pathlib.Path, default=None, help="File path to image be upsampled.")),
    (("-T", "--max_T"), dict(type=int, default=1_000, help="Max T value for noise scheduling (ensembles).")),
    # extension (not in the reference): bf16 storage of activations / contraction weights in the U-Net forward
    (("--bf16",), dict(action="store_true", help="Run the U-Net in bf16 storage (fp32 accumulation, sampler in fp32).")),
]


def generate_sr_images_diffusion(raw_args=None, lr_img=None, log=print, save_locally=True):
    parser = argparse.ArgumentParser(description="Generate Super Resolution Images.")
    for names, kw in _FLAGS:
        parser.add_argument(*names, **kw)
    args = vars(parser.parse_args(raw_args))
    if args["seed"] is not None:
        torch.manual_seed(args["seed"])
    if args["dest_path"] is not None and not args["dest_path"].exists():
        raise ValueError("Invalid destination path, kindly correct and ensure it exists!")
    out_dir = "./" if args["dest_path"] is None else str(args["dest_path"])
    if args["cold_step_size"] < 0 or args["cold_step_size"] > args["max_T"]:
        raise ValueError("Invalid step size for Cold Diffusion!")
    if lr_img is not None:
        if not isinstance(lr_img, np.ndarray):
            raise ValueError("Invalid low resolution image passed!")
    else:
        path = args["lr_img_path"]
        if path is None or not os.path.isfile(path) or str(path).rsplit(".", 1)[-1].lower() not in ("jpeg", "jpg", "png"):
            raise ValueError("Invalid/No path for low resolution image or unsupported image.")
        from PIL import Image
        lr_img = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]             # BGR like cv2.imread
    if args["device"] != "cuda":
        raise RuntimeError("This build runs the U-Net on MI355X only: pass --device cuda")
    dev = torch.device("cuda")
    lr_img = torch.from_numpy(((lr_img.astype(float) - 127.5) / 127.5)).float().permute(2, 0, 1).unsqueeze(0).to(dev)

    with open(args["config"], "r") as f:
        details = json.load(f)
    if "models" not in details or len(details["models"]) == 0:
        raise ValueError("Invalid / No model details in json, kindly correct and try again!")
    folder = os.path.split(args["config"])[0]

    noise = None
    for md in details["models"]:
        name = md["noise_scheduler"].lower()
        if name == NoiseScheduler.LINEAR.name.lower():
            schedule = NoiseDegradation(md["beta_1"], md["beta_T"], args["max_T"], dev)
        elif name == NoiseScheduler.COSINE.name.lower():
            schedule = CosineNoiseDegradation(args["max_T"])
        else:
            raise ValueError("Invalid noise scheduler type.")
        if noise is None:
            H, W = md["img_H"], md["img_W"]
            if H < lr_img.shape[2] or W < lr_img.shape[3]:
                raise ValueError("Invalid shapes for High Resolution and Low Resolution images.")
            noise = torch.randn((lr_img.shape[0], md["img_C"], H, W), device=dev)
            x_t = 1 * noise
            upsampled = ops.area_resize(lr_img, (H, W))
            cond = schedule(upsampled, torch.tensor([md["cond_t"]], device=dev), noise)
        else:                                      # ensemble: continue from the previous model's estimate
            x_t = schedule(delta, torch.tensor([md["max_noise"]], device=dev), noise)
        labels = None
        if md["cond_dim"] is not None:
            if args["labels"] is None or len(args["labels"]) != md["cond_dim"]:
                raise ValueError("Invalid/No conditional labels passed!")
            labels = torch.tensor(args["labels"]).float().to(dev)
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
        delta = cold_diffusion_sampling(diffusion_net=net, noise_degradation=schedule, x_t=x_t, noise=noise,
                                        min_noise=md["min_noise"], max_noise=md["max_noise"], cond_img=cond,
                                        labels_tensor=labels, skip_step_size=args["cold_step_size"], device=dev, log=log)
    result = upsampled + delta
    if not save_locally:
        return result
    name = f"{datetime.now().strftime('%d-%m-%Y %H:%M:%S')}_({H},{W})_{uuid.uuid4().hex}"
    plot_sampled_images(sampled_imgs=result, file_name=name, dest_path=out_dir)
    return None


if __name__ == "__main__":
    generate_sr_images_diffusion()
