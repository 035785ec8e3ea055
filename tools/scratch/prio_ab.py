"""A/B: HIP stream priorities for the training step (main compute stream vs the weight-gradient side stream vs the optimizer
stream).  python tools/scratch/prio_ab.py {c2|c4} {f32|bf16}"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
sys.path.insert(0, ROOT)
from bench import STANDARD  # noqa: E402
import degraders  # noqa: E402
from models.U_Net import U_Net  # noqa: E402
from sdm_hip import ops  # noqa: E402
from sdm_hip.trainer import DiffusionTrainer  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
res, n = {"c2": (64, 4), "c4": (128, 32)}[which]
dev = torch.device("cuda:0")
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())


def run(main_prio, side_prio, opt_prio, steps=40):
    torch.manual_seed(0)
    net = U_Net(**STANDARD).to(dev)
    ops.SIDE.streams.clear()
    ops.SIDE.streams[dev] = torch.cuda.Stream(device=dev, priority=side_prio)
    tr = DiffusionTrainer(net, degraders.CosineNoiseDegradation(1000), lr=2e-5, seed=1,
                          train_dtype=torch.bfloat16 if dtype == "bf16" else None)
    tr.opt_stream = torch.cuda.Stream(device=dev, priority=opt_prio)
    main = torch.cuda.Stream(device=dev, priority=main_prio) if main_prio is not None else torch.cuda.current_stream(dev)
    x0 = torch.randn(n, 3, res, res, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        for _ in range(3):
            tr.step(x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(x0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(f"{which} {dtype}: main {main_prio} side {side_prio} opt {opt_prio}: {dt*1e3:.3f} ms/step (loss {float(loss):.4f})", flush=True)
    del tr, net


lo, hi = torch.cuda.Stream.priority_range()
mode = sys.argv[3] if len(sys.argv) > 3 else "default"        # one configuration per process (streams map onto few hardware queues)
run(*{"default": (None, 0, 0), "main_hi": (hi, 0, 0), "main_opt_hi": (hi, 0, hi), "opt_hi": (None, 0, hi), "side_hi": (None, hi, 0)}[mode])
