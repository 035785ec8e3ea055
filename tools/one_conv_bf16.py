"""Run one bf16 conv shape repeatedly (profiling aid): python tools/one_conv_bf16.py n h w cin cout kind reps"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops
from models.custom_layers import TapConv, RunCtx
n, h, w, cin, cout = map(int, sys.argv[1:6]); kind = sys.argv[6]; reps = int(sys.argv[7])
conv = TapConv(cin, cout, kind).cuda()
wt = conv.taps_for(RunCtx(False, bf16=True))
x = ops.Act(torch.randn(n*h*w, cin, device="cuda").to(torch.bfloat16), n, h, w)
for _ in range(reps): y = ops.conv_fwd(x, wt, conv.bias, kind)
torch.cuda.synchronize()
print("done")
