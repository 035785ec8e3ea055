"""I/O helpers with the reference's names and file formats (utils/utils.py:8-95): progress bar, sample grid,
checkpoint save/load.  Nothing here is on the hot path."""
import os

import torch


def printProgressBar(iteration, total, prefix='', suffix='', decimals=1, length=100, fill='█', printEnd="\r", log=print):
    total = max(float(total), 1e-9)
    pct = f"{100 * (iteration / total):.{decimals}f}"
    filled = int(length * iteration // total)
    log(f"\r{prefix} |{fill * filled}{'-' * (length - filled)}| {pct}% {suffix}", end=printEnd)
    if iteration >= total:
        log()


def _grid(imgs, nrow=5, pad=2):
    """[N,3,H,W] in [-1,1] -> uint8 HxWx3 grid (what torchvision.utils.make_grid(normalize=True) produces)."""
    imgs = ((imgs.detach().float().cpu().clamp(-1, 1) + 1) * 0.5)
    n, c, h, w = imgs.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    canvas = torch.zeros(c, rows * (h + pad) + pad, cols * (w + pad) + pad)
    for i in range(n):
        r, q = divmod(i, cols)
        canvas[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = imgs[i]
    return (canvas * 255 + 0.5).clamp(0, 255).byte().permute(1, 2, 0).numpy()


def plot_sampled_images(sampled_imgs, file_name, dest_path=None, log=print):
    sampled_imgs = sampled_imgs[:, [2, 1, 0]]                      # images are BGR end-to-end in the reference
    dir_path = os.path.join(dest_path if dest_path is not None else os.path.dirname(os.path.abspath(__file__)), "plots")
    os.makedirs(dir_path, exist_ok=True)
    try:
        from PIL import Image
        path = os.path.join(dir_path, str(file_name) + ".jpg")
        Image.fromarray(_grid(sampled_imgs)).save(path)
        log(f"Saving generated image: {path}")
    except Exception as e:  # same forgiving behaviour as the reference
        log(f"An error occured while plotting reconstructed image: {e}")


def save_model(model_net, file_name, dest_path, checkpoint=False, steps=0, log=print):
    try:
        f_path = os.path.join(dest_path, "checkpoint" if checkpoint else "models")
        os.makedirs(f_path, exist_ok=True)
        torch.save(model_net, os.path.join(f_path, f"{file_name}_{str(steps)}.pt"))
        return True
    except Exception as e:
        log(f"Exception occured while saving model: {e}.")
        return False


def load_checkpoint(checkpoint_path, log=print):
    if not os.path.exists(checkpoint_path):
        log("Checkpoint does not exist.")
        return False, None
    log(f"Loading checkpoint: {checkpoint_path}")
    try:
        return True, torch.load(checkpoint_path, map_location=torch.device("cpu"), weights_only=True)
    except Exception:
        return False, None
