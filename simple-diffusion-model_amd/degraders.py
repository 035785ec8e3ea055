"""Noise schedules and the forward diffusion q(x_t | x_0) — drop-in for the reference's degraders.py
(NoiseDegradation :10-59, CosineNoiseDegradation :63-104).

Both schedules are tabulated once on the host with the reference's own float32 formulas (so the table
entries are bit-identical to what the reference computes per call on CPU) and kept on the device; the
per-image mix  sqrt(abar) x0 + sqrt(1-abar) eps  is one fused kernel (sdm_q_sample).
"""
import math

import torch
import torch.nn as nn

from sdm_hip import ops


class _TabulatedSchedule(nn.Module):
    """beta/alpha/alpha_bar tables for integer steps lo..T, indexed by step - lo."""

    def _install(self, lo, beta, alpha, abar):
        self._lo = lo
        # master tables stay on the host; per-device copies are made once and never moved or freed (a captured
        # hipGraph may hold their addresses)
        self.beta, self.alpha, self.alpha_cumulative_prod = beta.float(), alpha.float(), abar.float()
        self._per_device = {}

    def _tables(self, device):
        device = torch.device(device)
        if device.type == "cpu":
            return self.beta, self.alpha, self.alpha_cumulative_prod
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        t = self._per_device.get(device)
        if t is None:
            t = self._per_device[device] = tuple(x.to(device) for x in (self.beta, self.alpha, self.alpha_cumulative_prod))
        return t

    def get_timestep_params(self, step):
        b, a, ab = self._tables(step.device)
        i = step.to(torch.int64) - self._lo
        return b[i], a[i], ab[i]

    def alpha_bar_of(self, steps):
        _, _, ab = self._tables(steps.device)
        return ab[steps.to(torch.int64) - self._lo].contiguous()

    def host_params(self, step):
        """(beta, alpha, alpha_bar) of one integer step as python floats (sampler coefficient tables)."""
        i = int(step) - self._lo
        return float(self.beta[i]), float(self.alpha[i]), float(self.alpha_cumulative_prod[i])

    def forward(self, img, steps, eps=None):
        if not img.is_cuda:
            raise RuntimeError("noise degradation runs on the MI355X: pass CUDA tensors")
        if eps is None:
            eps = torch.randn_like(img)
        abar = self.alpha_bar_of(steps.to(img.device))
        if abar.numel() == 1 and img.shape[0] != 1:
            abar = abar.expand(img.shape[0]).contiguous()
        return ops.q_sample(img.float().contiguous(), eps.float().contiguous(), abar)


class NoiseDegradation(_TabulatedSchedule):
    """Linear schedule of DDPM: beta = linspace(beta_1, beta_T, T+1), indexed directly by t."""

    def __init__(self, beta_1, beta_T, max_noise_step, device="cpu"):
        super().__init__()
        self.beta_1, self.beta_T, self.max_noise_step = beta_1, beta_T, max_noise_step
        beta = torch.linspace(start=beta_1, end=beta_T, steps=int(max_noise_step + 1))
        alpha = 1 - beta
        self._install(0, beta, alpha, torch.cumprod(alpha, dim=0))


class CosineNoiseDegradation(_TabulatedSchedule):
    """Cosine schedule of Improved DDPM; beta clipped to [0.001, 0.999]."""

    def __init__(self, max_noise_step=1000):
        super().__init__()
        self.max_noise_step = max_noise_step
        self.offset = 0.008
        steps = torch.arange(-1, int(max_noise_step) + 1)          # integer steps, like the reference's callers pass
        abar = self.compute_alpha_bar(steps)
        prev = self.compute_alpha_bar(steps - 1)
        beta = torch.clip(1 - (abar / prev), min=0.001, max=0.999)
        self._install(-1, beta, 1 - beta, abar)

    def compute_alpha_bar(self, steps):
        f = lambda s: torch.cos((((s / self.max_noise_step) + self.offset) / (1 + self.offset)) * (math.pi / 2)) ** 2
        return f(steps) / f(torch.zeros_like(steps))
