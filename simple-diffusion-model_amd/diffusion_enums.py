"""Names kept from the reference's diffusion_enums.py (DiffusionAlg / NoiseScheduler)."""
from enum import Enum


class DiffusionAlg(Enum):
    DDPM = 0
    DDIM = 1


class NoiseScheduler(Enum):
    LINEAR = 0
    COSINE = 1
