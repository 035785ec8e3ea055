"""DDPM sampling-step timing: hipGraph replay vs eager, host enqueue vs GPU time (tuning aid)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from models.U_Net import U_Net
import degraders, diffusion_sampling_algorithms as S
torch.manual_seed(0)
net = U_Net(num_resnet_blocks=1, num_layers=4, attn_layers=[2, 3], time_dim=512, min_channel=128, max_channel=512).cuda().eval()
sched = degraders.CosineNoiseDegradation(1000)
quiet = lambda *a, **k: None
for n, dt in ((1, torch.float32), (4, torch.float32), (1, torch.bfloat16), (4, torch.bfloat16)):
    net.set_inference_dtype(dt)
    x = torch.randn(n, 3, 64, 64, device="cuda")
    print(dt)
    for use_graph in (True, False):
        S.ddpm_sampling(net, sched, x, 1, 5, device="cuda", log=quiet, use_graph=use_graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S.ddpm_sampling(net, sched, x, 1, 200, device="cuda", log=quiet, use_graph=use_graph)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"batch {n} graph={use_graph}: {200/(t2-t0):.1f} steps/s  ({(t2-t0)/200*1e3:.3f} ms/step, host returned after {(t1-t0)/200*1e3:.3f} ms/step)")
