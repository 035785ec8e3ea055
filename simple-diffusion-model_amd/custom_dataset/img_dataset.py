"""Image datasets with the reference's conventions (custom_dataset/img_dataset.py, conditional_img_dataset.py):
float CHW tensors in BGR channel order scaled to [-1, 1].  cv2 / TinyDB are not required: images are decoded
with PIL, the labelled variant reads the same TinyDB JSON file ({"_default": {"1": {"filepath":..., "labels": {...}}}})
with the json module.  ``synthetic:N:C:H:W`` yields seeded Gaussian images (benchmarks, smoke tests)."""
import glob
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def _load_bgr(path):
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    return torch.from_numpy(((rgb[:, :, ::-1] - 127.5) / 127.5).copy()).permute(2, 0, 1).contiguous()


class SyntheticDataset(Dataset):
    def __init__(self, spec, cond_dim=None):
        _, n, c, h, w = spec.split(":")
        self.shape, self.n, self.cond_dim = (int(c), int(h), int(w)), int(n), cond_dim

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(1234 + i)
        img = torch.randn(self.shape, generator=g)
        if self.cond_dim is None:
            return img
        return img, torch.eye(self.cond_dim)[i % self.cond_dim]

    def get_labels(self):
        return [f"class_{i}" for i in range(self.cond_dim or 0)]


class ImageDataset(Dataset):
    def __init__(self, img_paths):
        self.img_paths = list(img_paths)

    def __len__(self):
        return len(self.img_paths)

    def __getitem__(self, i):
        return _load_bgr(self.img_paths[i])


class ConditionalImgDataset(Dataset):
    def __init__(self, dataset_path):
        with open(dataset_path) as f:
            table = json.load(f).get("_default", {})
        self.records = [table[k] for k in sorted(table, key=lambda s: int(s))]
        self.labels = list(self.records[0]["labels"].keys()) if self.records else []
        self.root = os.path.dirname(os.path.abspath(dataset_path))

    def get_labels(self):
        return self.labels

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        r = self.records[i]
        path = r["filepath"] if os.path.isabs(r["filepath"]) else os.path.join(self.root, r["filepath"])
        return _load_bgr(path), torch.tensor([float(r["labels"][k]) for k in self.labels])


def open_dataset(dataset_path, use_conditional, cond_dim=None):
    if isinstance(dataset_path, str) and dataset_path.startswith("synthetic:"):
        return SyntheticDataset(dataset_path, cond_dim if use_conditional else None)
    if use_conditional:
        return ConditionalImgDataset(dataset_path)
    paths = sorted(glob.glob(dataset_path))
    if not paths:
        raise Exception("No dataset found!")
    return ImageDataset(paths)
