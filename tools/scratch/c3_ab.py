"""C3 (128x128, cond, bf16, DDIM + CFG, 64 images) step time under the round-3 switches."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd")); sys.path.insert(0, ROOT)
from bench import STANDARD
import degraders, diffusion_sampling_algorithms as S
from models.U_Net import U_Net
from sdm_hip import ops
torch.manual_seed(0)
net = U_Net(**dict(STANDARD, cond_dim=10)).cuda().eval()
net.set_inference_dtype(torch.bfloat16)
sched = degraders.CosineNoiseDegradation(1000)
n = 64
xT = torch.randn(n, 3, 128, 128, device="cuda")
lab = torch.nn.functional.one_hot(torch.randint(0, 10, (n,)), 10).float().cuda()
quiet = lambda *a, **k: None
def run():
    S.ddim_sampling(net, sched, xT, 1, 1000, None, lab, 500, "cuda", quiet, cfg_scale=3.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S.ddim_sampling(net, sched, xT, 1, 1000, None, lab, 40, "cuda", quiet, cfg_scale=3.0)      # 26 steps
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 26 * 1e3
print("default", round(run(), 2))
for name, obj in (("SMALL_ATTN", ops.SMALL_ATTN), ("TWO_PASS_ATTN", ops.TWO_PASS_ATTN), ("MOD_TABLE", S.MOD_TABLE), ("GN_COOP", ops.GN_COOP), ("NOGRAD_ATTN", ops.NOGRAD_ATTN)):
    obj[0] = False
    print(name, "off:", round(run(), 2), flush=True)
    obj[0] = True
