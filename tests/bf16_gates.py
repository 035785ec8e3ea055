"""Derived gates of the bf16 paths: the HIP bf16 forward / gradients against the oracle's bf16-STORAGE
mode (oracle/unet_oracle.py: Bf16Storage -- the reference's formulas with a round-to-bf16 exactly where the HIP path stores
bf16 and fp32 wherever it keeps fp32).

What the tolerance is, and why it is not "summation-order noise x rounding count" (~1e-4).  A bf16 store is a
discontinuous function of the fp32 value in front of it.  Two correct implementations differ in that value by fp32
evaluation-order noise (a few 2^-24 relative); wherever it sits that close to a rounding boundary the stored element
differs by a whole bf16 ulp (2^-8), the next layer sees a ~1e-4 perturbation, further roundings flip, and after a handful of
layers the two computations carry INDEPENDENT rounding noise.  The distance of a correct implementation to the storage
model is therefore the distance of the model to its own twin with four fp32 ulps of jitter in front of every store
(Bf16Storage.floor_twin) -- the FLOOR -- and it is as large as the bf16-vs-fp32 distance itself.  Measured on the GPU
(tools/bf16_oracle_probe.py, gpurun_out/r4_bf16_oracle_probe.log, round 4): HIP-vs-model / FLOOR = 0.97 .. 1.00 on whole-net
forwards; per gradient tensor median 0.97 .. 1.00, p90 1.12 .. 1.16, worst 1.94 (a 32-element GroupNorm bias); a net with
no normalisation (golden case no_emb), where nothing cascades, agrees with the model to 7e-8.
The floor is all-or-nothing per draw: a first flip needs an element within the jitter of a rounding boundary (probability
~ jitter / 2^-9 per stored element), and on the smallest golden net (sr_recon, 16x16) the default 2^-22 twin happens to flip
nothing while the HIP path's own noise does; once one element flips the cascade saturates at a level that does not depend on
the jitter.  The tests therefore draw the twin with jitter 2^-16 (still 256x below bf16 resolution) to get the saturated
floor, and bound it from below by half the model's own bf16 error (the saturated floor measured 0.8 .. 1.0 of it).
Gates:
  (1) rel-L2(HIP, model) <= K x max(FLOOR, 0.5 x rel-L2(model, fp32))   K = 3 forward, 3.5 per gradient tensor
      (the second term also carries the configurations where only one extra oracle pass is affordable: C3 / C4 / C5);
  (2) two-sided on the SIZE of the bf16 error: 0.4 <= rel-L2(HIP, fp32) / rel-L2(model, fp32) <= 2.5 on whole-net outputs
      (a path that silently ran fp32, or rounds where the model does not, fails here);
the fp32-oracle gates the tests already had (1e-2 forward, 2e-2 .. 4e-2 per gradient tensor) stay as the outer bound.
"""
import torch

K_FWD, K_GRAD = 3.0, 3.5
TWIN_JITTER = 2.0 ** -16


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_forward(y_hip, y_model, y_f32, y_twin=None, what=""):
    """y_model / y_twin / y_f32: oracle outputs in bf16-storage mode, its jittered twin (optional) and fp32."""
    e_model = rel(y_model, y_f32)
    floor = max(rel(y_twin, y_model) if y_twin is not None else 0.0, 0.5 * e_model)
    d = rel(y_hip, y_model)
    e_hip = rel(y_hip, y_f32)
    print(f"bf16 gate {what}: HIP vs storage model {d:.2e} (floor {floor:.2e}); bf16 error HIP {e_hip:.2e} / model {e_model:.2e}")
    assert d <= K_FWD * floor, (what, d, floor)
    assert 0.4 * e_model <= e_hip <= 2.5 * e_model, (what, e_hip, e_model)
    return d, floor


def check_grads(named_hip, g_model, g_f32, g_twin=None, skip=lambda k: False, what=""):
    """named_hip: {name: HIP gradient}; g_*: {name: oracle gradient} in the three modes.  Returns the number checked."""
    worst, n = (0.0, None), 0
    bad = []
    for k, g in named_hip.items():
        if skip(k) or g_f32.get(k) is None:
            continue
        e_model = rel(g_model[k], g_f32[k])
        floor = max(rel(g_twin[k], g_model[k]) if g_twin is not None else 0.0, 0.5 * e_model)
        d = rel(g, g_model[k])
        n += 1
        if d / max(floor, 1e-30) > worst[0]:
            worst = (d / max(floor, 1e-30), k)
        if d > K_GRAD * floor:
            bad.append((k, d, floor))
    print(f"bf16 gate {what}: {n} gradient tensors, worst HIP-vs-model / floor = {worst[0]:.2f} ({worst[1]})")
    assert not bad, bad[:8]
    return n


def oracle_grads(O, P, cfg, x, t, cond, target, storage):
    """(loss, {name: grad}) of mse(unet_forward(x), target) through the oracle in the given storage mode."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = O.unet_forward(leaves, cfg, x, t, cond, storage=storage)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in leaves.items()}
