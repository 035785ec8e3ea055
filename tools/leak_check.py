"""300 training steps and 3 sampling runs of the bench workload: device memory must be flat after warm-up."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
import diffusion_sampling_algorithms as S  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402


def main():
    torch.manual_seed(0)
    net = U_Net(**STANDARD).cuda()
    sched = degraders.CosineNoiseDegradation(1000)
    tr = DiffusionTrainer(net, sched, lr=2e-5, seed=1)
    x0 = torch.randn(4, 3, 64, 64, device="cuda")
    marks = []
    for i in range(300):
        loss = tr.step(x0)
        if i % 50 == 49:
            torch.cuda.synchronize()
            marks.append((i + 1, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, float(loss)))
            print("step %4d  allocated %6d MiB  reserved %6d MiB  loss %.4f" % marks[-1], flush=True)
    assert marks[-1][1] <= marks[1][1] + 8, "allocated memory grows during training"
    assert all(m[3] == m[3] for m in marks), "NaN loss"
    net.eval()
    quiet = lambda *a, **k: None
    for dt in (torch.float32, torch.bfloat16, torch.float32):
        net.set_inference_dtype(dt)
        S.ddpm_sampling(net, sched, torch.randn(4, 3, 64, 64, device="cuda"), 1, 40, device="cuda", log=quiet)
        torch.cuda.synchronize()
        print(str(dt), "sampling done: allocated %d MiB reserved %d MiB" % (torch.cuda.memory_allocated() >> 20,
                                                                         torch.cuda.memory_reserved() >> 20), flush=True)
    print("ok")


if __name__ == "__main__":
    main()
