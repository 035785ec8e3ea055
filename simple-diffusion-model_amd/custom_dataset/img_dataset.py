"""Image datasets with the reference's conventions (custom_dataset/img_dataset.py, conditional_img_dataset.py):
float CHW tensors in BGR channel order scaled to [-1, 1].  cv2 / TinyDB are not required: images are decoded
with PIL, the labelled variant reads the reference's TinyDB JSON file with the json module -- table "Data" holds one
record per image ({"filename": ..., "<label>": value, ...}), table "Labels" one record {"labels": [names]}
(conditional_img_dataset.py:19-36).  ``synthetic:N:C:H:W`` yields seeded Gaussian images (benchmarks, smoke tests)."""
import glob
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def _load_bgr_u8(path):
    """Decoded image as cv2.imread returns it: uint8 [H, W, 3], BGR."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)[:, :, ::-1])


def _shuffle_records(records):
    """The reference shuffles the records once, unseeded, at construction (conditional_img_dataset.py:33,
    doodle_dataset.py:38): harmless there, it has one process.  Under data parallelism (WORLD_SIZE > 1) every rank builds
    its own dataset object, and DistributedSampler / CachedLoader hand out INDEX shards -- disjoint only if all ranks hold
    the same record order -- so there the shuffle is seeded identically on every rank (SDM_DATA_SEED, default
    a constant); a single process keeps the reference's plain random.shuffle."""
    import random
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        random.Random(int(os.environ.get("SDM_DATA_SEED", "20240229"))).shuffle(records)
    else:
        random.shuffle(records)


def _load_bgr(path):
    """img_dataset.py:26-36: (img.astype(float) - 127.5) / 127.5 in float64, then .float(), then CHW."""
    bgr = _load_bgr_u8(path)
    return torch.from_numpy((bgr.astype(float) - 127.5) / 127.5).float().permute(2, 0, 1).contiguous()


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
    """TinyDB file of the reference (custom_dataset/conditional_img_dataset.py:14-36): tables "Data" and "Labels"; same
    exceptions when either is empty; the initial shuffle of the records is kept (seedable through ``random``)."""

    def __init__(self, dataset_path=None):
        with open(dataset_path) as f:
            db = json.load(f)
        data_tbl, labels_tbl = db.get("Data", {}), db.get("Labels", {})
        if len(data_tbl) <= 0:
            raise Exception("No data found in Data table.")
        if len(labels_tbl) <= 0:
            raise Exception("No data found in Labels table.")
        by_id = lambda tbl: [tbl[k] for k in sorted(tbl, key=lambda s: int(s))]      # TinyDB.all() order: document id
        self.all_labels = by_id(labels_tbl)[0]["labels"]
        self.all_data = by_id(data_tbl)
        _shuffle_records(self.all_data)
        self.root = os.path.dirname(os.path.abspath(dataset_path))
        self.dataset = [(d["filename"], [float(d[label]) for label in self.all_labels]) for d in self.all_data]

    def get_labels(self):
        return self.all_labels

    def __len__(self):
        return len(self.dataset)

    def _path(self, path):
        return path if (os.path.isabs(path) or os.path.exists(path)) else os.path.join(self.root, path)

    def __getitem__(self, i):
        if len(self.dataset) <= 0:
            raise Exception("No data found in dataset.")
        path, labels = self.dataset[i]
        return _load_bgr(self._path(path)), torch.Tensor(labels)


class DoodleImgDataset(Dataset):
    """(image, label image) pairs (custom_dataset/doodle_dataset.py:14-66): the same TinyDB file layout as the conditional
    dataset, the FIRST entry of the "Labels" list naming the key that holds the label image's path."""

    def __init__(self, dataset_path=None):
        with open(dataset_path) as f:
            db = json.load(f)
        data_tbl, labels_tbl = db.get("Data", {}), db.get("Labels", {})
        if len(data_tbl) <= 0:
            raise Exception("No data found in Data table.")
        if len(labels_tbl) <= 0:
            raise Exception("No data found in Labels table.")
        by_id = lambda tbl: [tbl[k] for k in sorted(tbl, key=lambda s: int(s))]
        self.all_labels = by_id(labels_tbl)[0]["labels"]
        self.all_data = by_id(data_tbl)
        _shuffle_records(self.all_data)
        self.root = os.path.dirname(os.path.abspath(dataset_path))
        label = self.all_labels[0]
        self.dataset = [(d["filename"], d[label]) for d in self.all_data]

    def get_labels(self):
        return self.all_labels

    def __len__(self):
        return len(self.dataset)

    def _path(self, p):
        return p if (os.path.isabs(p) or os.path.exists(p)) else os.path.join(self.root, p)

    def __getitem__(self, i):
        if len(self.dataset) <= 0:
            raise Exception("No data found in dataset.")
        img, lab = self.dataset[i]
        return _load_bgr(self._path(img)), _load_bgr(self._path(lab))


class SyntheticPairDataset(Dataset):
    """``synthetic-pairs:N:C:H:W``: seeded Gaussian (image, label image) pairs (doodle smoke tests)."""

    def __init__(self, spec):
        _, n, c, h, w = spec.split(":")
        self.shape, self.n = (int(c), int(h), int(w)), int(n)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(4321 + i)
        return torch.randn(self.shape, generator=g), torch.randn(self.shape, generator=g)

    def get_labels(self):
        return ["label_img"]


def open_pair_dataset(dataset_path):
    if isinstance(dataset_path, str) and dataset_path.startswith("synthetic-pairs:"):
        return SyntheticPairDataset(dataset_path)
    return DoodleImgDataset(dataset_path)


def open_dataset(dataset_path, use_conditional, cond_dim=None):
    if isinstance(dataset_path, str) and dataset_path.startswith("synthetic:"):
        return SyntheticDataset(dataset_path, cond_dim if use_conditional else None)
    if use_conditional:
        return ConditionalImgDataset(dataset_path)
    paths = sorted(glob.glob(dataset_path))
    if not paths:
        raise Exception("No dataset found!")
    return ImageDataset(paths)
