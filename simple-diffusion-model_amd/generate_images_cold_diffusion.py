"""Sample images with the cold-diffusion (x0-prediction) sampler on MI355X — same command line and inference
``config.json`` schema as the reference's generate_images_cold_diffusion.py (:33-79 flags, :120-196 "models" list: the
first model starts from x_T = noise, each further one from q(x_t | x0_hat, max_noise) with the SAME noise tensor).
``--device cpu`` is accepted by the parser for compatibility but refused at run time."""
import argparse
import json
import os
import pathlib
import uuid
from datetime import datetime

import torch

from degraders import CosineNoiseDegradation, NoiseDegradation
from diffusion_enums import NoiseScheduler
from diffusion_sampling_algorithms import cold_diffusion_sampling
from generate_images_diffusion import _UNET_KEYS
from models.U_Net import U_Net
from utils.utils import load_checkpoint, plot_sampled_images


def generate_images_cold_diffusion(raw_args=None, log=print, save_locally=True):
    parser = argparse.ArgumentParser(description="Generate Images using Cold Diffusion models.")
    parser.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu",
                        help="Hardware device (the kernels need 'cuda').")
    parser.add_argument("-c", "--config", required=True, type=pathlib.Path, help="File path to load config file.")
    parser.add_argument("-s", "--seed", type=int, default=None, help="Seed value for generating image (default: None).")
    parser.add_argument("-T", "--max_T", type=int, default=1_000, help="Max T value for noise scheduling (ensembles).")
    parser.add_argument("-n", "--num_images", type=int, default=1, help="Number of images to generate (default=1).")
    parser.add_argument("-d", "--dest_path", type=pathlib.Path, help="Folder for the generated grid (default: ./).")
    parser.add_argument("--cold_step_size", type=int, default=10, help="Number of steps to skip when using cold diffusion.")
    parser.add_argument("-l", "--labels", nargs="*", type=float, default=None, help="Conditional Labels.")
    parser.add_argument("--bf16", action="store_true", help="Run the U-Net in bf16 storage (extension).")
    args = vars(parser.parse_args(raw_args))

    if args["seed"] is not None:
        torch.manual_seed(args["seed"])
    if args["num_images"] <= 0:
        raise ValueError("Invalid image numbers, should be greater than 0!")
    if args["dest_path"] is not None and not args["dest_path"].exists():
        raise ValueError("Invalid destination path!")
    out_dir = "./" if args["dest_path"] is None else str(args["dest_path"])
    if args["cold_step_size"] < 0 or args["cold_step_size"] > args["max_T"]:
        raise ValueError("Invalid step size for Cold Diffusion!")
    if args["device"] != "cuda":
        raise RuntimeError("This build runs the U-Net on MI355X only: pass --device cuda")
    dev = torch.device("cuda")

    with open(args["config"], "r") as f:
        details = json.load(f)
    if "models" not in details or len(details["models"]) == 0:
        raise ValueError("Invalid / No model details in json, kindly correct and try again!")
    folder = os.path.split(args["config"])[0]

    noise, x0_approx, shape = None, None, None
    for md in details["models"]:
        if md["noise_scheduler"] == NoiseScheduler.LINEAR.name:
            schedule = NoiseDegradation(md["beta_1"], md["beta_T"], args["max_T"], dev)
        elif md["noise_scheduler"] == NoiseScheduler.COSINE.name:
            schedule = CosineNoiseDegradation(args["max_T"])
        else:
            raise ValueError("Invalid noise scheduler type.")
        if noise is None:                                   # X_T ~ N(0, I)
            shape = (args["num_images"], md["img_C"], md["img_H"], md["img_W"])
            noise = torch.randn(shape, device=dev)
            x_t = 1 * noise
        else:                                               # ensembles: restart from the previous estimate
            x_t = schedule(x0_approx, torch.tensor([md["max_noise"]], device=dev), noise)
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
        x0_approx = cold_diffusion_sampling(diffusion_net=net, noise_degradation=schedule, x_t=x_t, noise=noise,
                                            min_noise=md["min_noise"], max_noise=md["max_noise"], cond_img=None,
                                            labels_tensor=labels, skip_step_size=args["cold_step_size"], device=dev, log=log)

    if not save_locally:
        return x0_approx
    name = f"{datetime.now().strftime('%d-%m-%Y %H:%M:%S')}({shape[2]},{shape[3]})_{uuid.uuid4().hex}"
    plot_sampled_images(sampled_imgs=x0_approx, file_name=name, dest_path=out_dir, log=log)
    return None


if __name__ == "__main__":
    generate_images_cold_diffusion()
