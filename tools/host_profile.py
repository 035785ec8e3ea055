"""cProfile of the host side of the training step (which Python calls the enqueue time goes to). MI355X only."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = U_Net(**STANDARD).to(dev)
    tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-5, seed=1)
    x0 = torch.randn(4, 3, 64, 64, device=dev)
    for _ in range(5):
        tr.step(x0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        tr.step(x0)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(40)
    st.sort_stats("cumulative").print_stats(50)


if __name__ == "__main__":
    main()
