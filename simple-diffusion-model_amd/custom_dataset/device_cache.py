"""Device-resident input pipeline.

The reference decodes every image again each epoch (cv2.imread in 4 DataLoader workers, img_dataset.py:26) and flips per
image in a Python lambda (train_diffusion.py:312-314); at a few milliseconds per training step that becomes the
bottleneck of an 8-GPU run.  Here the dataset is decoded ONCE into a uint8 [N, H, W, 3] tensor (BGR, exactly what cv2
returns) that lives in HBM -- 100k images of 128x128 are 4.9 GB of the 288 GB -- and a batch is one fused kernel
(sdm_u8_gather_nchw): gather by index, normalise with the reference's own (v - 127.5) / 127.5 table, horizontal flip,
HWC -> CHW.  Values are bit-identical to what the datasets in img_dataset.py return.  File formats are unchanged: a glob
of images, or the TinyDB JSON of the conditional dataset.
"""
import numpy as np
import torch

from sdm_hip import lib, ops

from .img_dataset import ConditionalImgDataset, ImageDataset, _load_bgr_u8


def norm_table():
    """(v - 127.5) / 127.5 for v = 0..255, float64 arithmetic rounded once to float32 (img_dataset.py:29-32)."""
    return torch.from_numpy(((np.arange(256).astype(float) - 127.5) / 127.5)).float()


class DeviceImageCache:
    def __init__(self, dataset, device, log=None):
        if isinstance(dataset, ImageDataset):
            paths, labels = dataset.img_paths, None
        elif isinstance(dataset, ConditionalImgDataset):
            paths = [dataset._path(p) for p, _ in dataset.dataset]
            labels = torch.tensor([lab for _, lab in dataset.dataset], dtype=torch.float32)
        else:
            raise TypeError("DeviceImageCache holds decoded image files (ImageDataset / ConditionalImgDataset)")
        first = _load_bgr_u8(paths[0])
        h, w, c = first.shape
        host = torch.empty((len(paths), h, w, c), dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else \
            torch.empty((len(paths), h, w, c), dtype=torch.uint8)
        for i, p in enumerate(paths):
            img = first if i == 0 else _load_bgr_u8(p)
            if img.shape != (h, w, c):
                raise ValueError(f"{p}: {img.shape}, expected {(h, w, c)} (all images of a dataset share one size)")
            host[i] = torch.from_numpy(img)
            if log is not None and (i + 1) % 10000 == 0:
                log(f"decoded {i + 1:,} / {len(paths):,} images")
        self.device = torch.device(device)
        self.images = host.to(self.device, non_blocking=True)
        self.labels = None if labels is None else labels.to(self.device)
        self.lut = norm_table().to(self.device)
        self.n, self.h, self.w, self.c = len(paths), h, w, c

    def __len__(self):
        return self.n

    def batch(self, idx, flip=None):
        """idx: int64 [B] on the device; flip: optional bool/uint8 [B].  -> (x0 fp32 [B, C, H, W], labels or None)."""
        idx = idx.to(self.device, torch.int64).contiguous()
        out = torch.empty((idx.numel(), self.c, self.h, self.w), dtype=torch.float32, device=self.device)
        f = None if flip is None else flip.to(self.device, torch.uint8).contiguous()
        lib.call("sdm_u8_gather_nchw", self.images.data_ptr(), idx.data_ptr(), None if f is None else f.data_ptr(),
                 self.lut.data_ptr(), out.data_ptr(), idx.numel(), self.c, self.h, self.w, ops._stream())
        return out, (None if self.labels is None else self.labels.index_select(0, idx))


class CachedLoader:
    """Epoch iterator over a DeviceImageCache: a fresh device-side permutation per epoch (per-rank generator stream),
    rank-strided shards under data parallelism (what DistributedSampler gives), optional p=0.5 horizontal flips.  Yields
    what the reference's DataLoader yields -- x0, or (x0, labels) -- already on the device."""

    def __init__(self, cache, batch_size, rank=0, world=1, flip=False, drop_last=False, generator=None, conditional=False):
        self.cache, self.bs, self.rank, self.world, self.flip = cache, batch_size, rank, world, flip
        self.drop_last = drop_last or world > 1
        self.gen = generator
        self.conditional = conditional
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_rank = len(self.cache) // self.world if self.world > 1 else len(self.cache)
        return per_rank // self.bs if self.drop_last else (per_rank + self.bs - 1) // self.bs

    def __iter__(self):
        dev = self.cache.device
        # the permutation must be the same on every rank: seeded by the epoch, not by the per-rank stream
        g = torch.Generator(device=dev).manual_seed(0x5D + self.epoch)
        perm = torch.randperm(len(self.cache), device=dev, generator=g)
        if self.world > 1:
            per = len(self.cache) // self.world
            perm = perm[self.rank:per * self.world:self.world]
        for b in range(len(self)):
            idx = perm[b * self.bs:(b + 1) * self.bs]
            flip = (torch.rand(idx.numel(), device=dev, generator=self.gen) < 0.5) if self.flip else None
            x0, labels = self.cache.batch(idx, flip)
            yield (x0, labels) if self.conditional else x0
