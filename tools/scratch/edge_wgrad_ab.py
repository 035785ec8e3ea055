"""fp32 edge-conv weight gradients (128 -> 3 head, 3 -> 128 stem) at 64x64 batch 4: the dedicated VALU kernels against the
matrix-core weight-gradient kernel on NHWC rows."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "simple-diffusion-model_amd"))
from sdm_hip import ops, lib
from sdm_hip.ops import Act
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
n, H, W, C = 4, 64, 64, 128
g = torch.Generator().manual_seed(0)
hin = torch.randn(n * H * W, C, generator=g).cuda()
dout = torch.randn(n, 3, H, W, generator=g).cuda()
# current kernels
dw = torch.zeros(9, 3, C, device="cuda"); db = torch.zeros(3, device="cuda")
need = lib.raw("sdm_edge_wgrad_workspace_bytes")(3, C) // 4
slab = torch.empty(need, device="cuda")
def cur():
    dw.zero_(); db.zero_()
    lib.call("sdm_head_conv_wgrad", hin.data_ptr(), dout.data_ptr(), dw.data_ptr(), db.data_ptr(), slab.data_ptr(), n, C, H, W, 3, ops._stream())
print("head wgrad, dedicated kernels:", round(t(cur), 1), "us")
ref = dw.clone()
for pad in (4, 8):
    rows = torch.zeros(n * H * W, pad, device="cuda")
    dw2 = torch.zeros(9, 3, C, device="cuda"); db2 = torch.zeros(3, device="cuda")
    def mm():
        rows[:, :3] = dout.permute(0, 2, 3, 1).reshape(-1, 3)
        dw2.zero_(); db2.zero_()
        ops.conv_wgrad(rows[:, :3], (H, W), Act(hin, n, H, W), "c3s1", dw2, dbias=db2)
    print(f"head wgrad, matrix cores (rows padded to {pad}) incl. permute + zero:", round(t(mm), 1), "us", float((dw2 - ref).norm() / ref.norm()))
    def mm_only():
        ops.conv_wgrad(rows[:, :3], (H, W), Act(hin, n, H, W), "c3s1", dw2, dbias=db2)
    print("   kernel only:", round(t(mm_only), 1), "us")
# stem: dW[9][128][3]
x = torch.randn(n, 3, H, W, generator=g).cuda()
dz = torch.randn(n * H * W, C, generator=g).cuda()
dws = torch.zeros(9, C, 3, device="cuda"); dbs = torch.zeros(C, device="cuda")
def cur_s():
    lib.call("sdm_stem_conv_wgrad", x.data_ptr(), dz.data_ptr(), dws.data_ptr(), dbs.data_ptr(), slab.data_ptr(), n, 3, H, W, C, ops._stream())
print("stem wgrad, dedicated kernels:", round(t(cur_s), 1), "us")
xr = torch.zeros(n * H * W, 4, device="cuda"); xr[:, :3] = x.permute(0, 2, 3, 1).reshape(-1, 3)
dws2 = torch.zeros(9, C, 3, device="cuda"); dbs2 = torch.zeros(C, device="cuda")
def mm_s():
    ops.conv_wgrad(dz, (H, W), Act(xr[:, :3], n, H, W), "c3s1", dws2, dbias=dbs2)
print("stem wgrad, matrix cores, kernel only:", round(t(mm_s), 1), "us")
