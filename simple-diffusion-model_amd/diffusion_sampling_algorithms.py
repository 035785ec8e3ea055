"""DDPM / DDIM / cold-diffusion samplers — drop-in for the reference's diffusion_sampling_algorithms.py
(same function names and arguments, :5-14, :66-76, :150-161).

The reference runs a Python loop of T iterations, each with a U-Net call, ~10 elementwise torch ops and a
host->device copy of t.  Here ONE denoise step — [step-table cursor -> U-Net forward -> fused update] — is
captured as a hipGraph and replayed T times: the per-step timestep and update coefficients are read from
device-resident tables through a cursor the graph itself advances, so a replay needs no host work.
Coefficient tables are computed with the reference's own float32 expressions.

Extra keyword-only arguments (not in the reference): ``noise`` injects the per-step z tensors (trajectory
parity tests; GPU and CPU generators differ), ``use_graph`` forces/disables hipGraph capture, ``cfg_scale`` turns on
classifier-free guidance (BASELINE config 3): the net runs once per step on [x | x] with [labels | 0] and the two
halves are blended as e_u + w (e_c - e_u).
"""
import os

import torch

from sdm_hip import lib, ops
from utils.utils import printProgressBar


def _use_graph(flag):
    if flag is not None:
        return flag
    return os.environ.get("SDM_HIPGRAPH", "1") != "0"


PLAN = [os.environ.get("SDM_PLAN", "0") == "1"]       # replay the step from the C-side launch plan instead of the hipGraph
                                                      # (opt-in: measured equal to graph replay, 1.078 vs 1.081 ms at N=1)
MOD_TABLE = [os.environ.get("SDM_MOD_TABLE", "1") != "0"]       # modulation tables per run (U_Net.modulation_tables)
MOD_TABLE_MAX_BYTES = 1 << 30


class _StepRunner:
    """Owns the static buffers of one sampling run and replays the captured step."""

    def __init__(self, net, x_t, t_steps, coef, cond_img, labels, noise, use_graph, cfg_scale=None, seed=0):
        dev = x_t.device
        self.net, self.cond_img, self.labels, self.noise = net, cond_img, labels, noise
        self.cfg_scale = cfg_scale
        ce = getattr(net, "cond_emb", None)
        if labels is None and ce is not None and getattr(ce, "cond_layer", None) is not None:
            # the reference fails here too (custom_layers.py:96-97 runs cond_layer(None)); the per-run modulation tables
            # would otherwise silently drop the conditional term
            raise ValueError("this U_Net has a label embedding (cond_dim): labels_tensor is required")
        if cfg_scale is not None:
            if labels is None:
                raise ValueError("classifier-free guidance needs labels")
            lab = labels if labels.dim() == 2 else labels[None, :].expand(x_t.shape[0], -1)
            self.labels2 = torch.cat((lab.float(), torch.zeros_like(lab, dtype=torch.float32)), dim=0).contiguous()
        self.x = x_t.detach().to(torch.float32).contiguous().clone()
        self.nsteps, self.ncoef = len(t_steps), coef.shape[1]
        self.t_table = torch.tensor(t_steps, dtype=torch.int64, device=dev)
        self.coef_table = coef.to(torch.float32).contiguous().to(dev)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)       # index of the step being run
        self.t_cur = torch.zeros(1, dtype=torch.int64, device=dev)
        self.coef_cur = torch.zeros(self.ncoef, dtype=torch.float32, device=dev)
        self.graph = None
        self.body = None
        self.use_graph = use_graph
        self.seed = seed      # Philox key of the DDPM update's in-kernel noise (ddpm_sampling draws it; 0 elsewhere: unused)
        # AdaGN modulation of every step of the run, tabulated once (the embedding depends on t and the labels only)
        self.mod_time = self.mod_cond = self.mods_cur = self.mod_views = None
        lab = self.labels2 if cfg_scale is not None else labels
        rows = 1 if (lab is None or lab.dim() == 1) else lab.shape[0]
        tables = getattr(net, "modulation_tables", None)
        if MOD_TABLE[0] and tables is not None and getattr(net, "cond_emb", None) is not None:
            width = sum((a.y_scale.weight.shape[0] + 3) // 4 * 4 for a in net._adagns)
            if width and 4 * width * (self.nsteps + 2 * rows) <= MOD_TABLE_MAX_BYTES:
                with torch.no_grad():
                    self.mod_time, self.mod_cond, cols = tables(self.t_table, lab)
                rows = 1 if self.mod_cond is None else self.mod_cond.shape[0]
                self.mods_cur = torch.empty((rows, width), dtype=torch.float32, device=dev)
                self.mod_views = [self.mods_cur[:, c0:c1] for c0, c1 in cols]

    def begin(self):
        """Head of the step: this step's t, update coefficients and modulation rows out of the per-run tables."""
        mt = self.mod_time
        lib.call("sdm_step_begin", self.cursor.data_ptr(), self.t_table.data_ptr(), self.coef_table.data_ptr(), self.ncoef,
                 self.nsteps, self.t_cur.data_ptr(), self.coef_cur.data_ptr(), None if mt is None else mt.data_ptr(),
                 None if self.mod_cond is None else self.mod_cond.data_ptr(),
                 None if mt is None else self.mods_cur.data_ptr(), 0 if mt is None else self.mods_cur.shape[0],
                 0 if mt is None else self.mods_cur.shape[1], ops._stream())

    def _net(self, x, t, labels):
        if self.mod_views is None:
            return self.net(x, t, labels)
        self.net._preset_mods = self.mod_views
        try:
            return self.net(x, t, labels)
        finally:
            self.net._preset_mods = None

    def eps(self):
        xin = self.x if self.cond_img is None else torch.cat((self.x, self.cond_img), dim=1)
        if self.cfg_scale is None:
            return self._net(xin, self.t_cur, self.labels)
        n = xin.shape[0]
        e2 = self._net(torch.cat((xin, xin), dim=0), self.t_cur.expand(2 * n).contiguous(), self.labels2)
        return ops.cfg_combine(e2, self.cfg_scale)

    def z(self):
        """Injected per-step noise (trajectory tests), else None: the DDPM update draws it in the kernel."""
        if self.noise is None:
            return None
        idx = self.cursor.clamp(max=self.nsteps - 1).to(torch.int64)      # the update kernel advances the cursor afterwards
        return self.noise.index_select(0, idx)[0]

    def run(self, body, on_step):
        if self.use_graph:
            x0 = self.x.clone()
            s = _warmup_stream(self.x.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                               # warm-up: sizes workspaces, sets kernel attributes
                body()
            torch.cuda.current_stream().wait_stream(s)
            self.x.copy_(x0)
            self.cursor.zero_()
            # SDM_PLAN=1: when the step consists of library launches only (no conditioning image to concatenate, no guidance
            # batch to build, no injected noise to index: those go through torch kernels), the capture is ALSO recorded as
            # a C-side launch plan (sdm_plan_*) and the plan is replayed instead of the graph: plain launches issued from
            # one C call per step.  Measured on MI355X: the same step time as graph replay (1.078 vs 1.081 ms at N=1 fp32;
            # eager launching 1.030 ms per step once the one-off capture, ~10 ms, is not amortised into it) -- so the
            # graph stays the default.  The graph stays alive either way: its memory pool owns every buffer the plan uses.
            planned = PLAN[0] and self.cond_img is None and self.cfg_scale is None and self.noise is None
            self.graph = torch.cuda.CUDAGraph()
            plan = 0
            with ops.capture_scope() as kept, torch.cuda.graph(self.graph):
                if planned:
                    lib.call("sdm_plan_begin")
                try:
                    body()
                finally:
                    if planned:
                        plan = lib.call("sdm_plan_end")              # 0: something in the step cannot be replayed by a plan
            self._kept = kept             # library buffers the graph points at live exactly as long as the graph
            self.x.copy_(x0)
            self.cursor.zero_()
            try:
                if plan:
                    st = ops._stream()
                    for i in range(self.nsteps):
                        lib.call("sdm_plan_run", plan, 1, st)
                        on_step(i)
                else:
                    for i in range(self.nsteps):
                        self.graph.replay()
                        on_step(i)
            finally:
                if plan:
                    lib.call("sdm_plan_free", plan)
        else:
            for i in range(self.nsteps):
                body()
                on_step(i)
        lib.check_device_error()      # a bounded device-side wait that gave up (cooperative GroupNorm) is an error, not a NaN image


_WARMUP_STREAMS = {}


def _warmup_stream(device):
    """ONE side stream per device for the warm-up step in front of a capture (a fresh stream per run would leave one set of
    per-stream scratch behind per run, ops._StreamKeyed)."""
    st = _WARMUP_STREAMS.get(device)
    if st is None:
        st = _WARMUP_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


def _require_cuda(x_t):
    if not x_t.is_cuda:
        raise RuntimeError("sampling runs on the MI355X: pass CUDA tensors / --device cuda")


def ddpm_sampling(diffusion_net, noise_degradation, x_t, min_noise=1, max_noise=1_000, cond_img=None,
                  labels_tensor=None, device="cpu", log=print, *, noise=None, use_graph=None, cfg_scale=None):
    _require_cuda(x_t)
    dev = x_t.device
    diffusion_net.eval()
    steps = list(range(max_noise, min_noise - 1, -1))
    beta, alpha, abar = noise_degradation.get_timestep_params(torch.tensor(steps))      # float32, host tables
    beta, alpha, abar = beta.cpu(), alpha.cpu(), abar.cpu()
    sigma = beta ** 0.5
    sigma = torch.where(torch.tensor(steps) > 1, sigma, torch.zeros_like(sigma))       # z = 0 when noise_step <= 1
    coef = torch.stack((1 / (alpha ** 0.5), (1 - alpha) / ((1 - abar) ** 0.5), sigma), dim=1)
    if cond_img is not None:
        cond_img = cond_img.to(dev)
    if labels_tensor is not None:
        labels_tensor = labels_tensor.to(dev)
    if noise is not None:
        noise = noise.to(dev).float().contiguous()
    # z ~ N(0, 1) of the update (reference :43, torch.randn_like on the device) is drawn IN the update kernel (Philox4x32-10,
    # Box-Muller); its key comes from the DEVICE generator, so torch.cuda.manual_seed / torch.manual_seed govern the
    # samples as they do in the reference (the stream itself differs from torch's: INTEGRATION.md).  Only this sampler
    # draws: the deterministic DDIM / cold samplers leave every generator untouched.
    seed = int(torch.randint(0, 2 ** 62, (1,), device=dev).item()) if noise is None else 0
    r = _StepRunner(diffusion_net, x_t, steps, coef, cond_img, labels_tensor, noise, _use_graph(use_graph), cfg_scale, seed=seed)

    def body():
        r.begin()
        e = r.eps()
        ops.ddpm_update(r.x, e, r.z(), r.coef_cur, seed=r.seed, rng_step=r.t_cur, cursor=r.cursor, nsteps=r.nsteps)

    def on_step(i):
        printProgressBar(iteration=i, total=max_noise - min_noise, prefix='Iterations:', suffix='Complete', length=50, log=log)

    with torch.no_grad():
        r.run(body, on_step)
    return r.x


def ddim_sampling(diffusion_net, noise_degradation, x_t, min_noise=1, max_noise=1_000, cond_img=None,
                  labels_tensor=None, ddim_step_size=10, device="cpu", log=print, *, use_graph=None, cfg_scale=None):
    _require_cuda(x_t)
    dev = x_t.device
    diffusion_net.eval()
    steps = list(range(max_noise, min_noise - 1, -ddim_step_size))
    if min_noise not in steps:
        steps = steps + [min_noise]
    abar = noise_degradation.get_timestep_params(torch.tensor(steps))[2].cpu()
    abar_next = torch.cat((abar[1:], abar[-1:]))                     # last entry unused (no update after the last step)
    upd = torch.ones_like(abar)
    upd[-1] = 0.0                                                    # the reference leaves x_t alone on the last step
    # eta = 0 (hard-coded in the reference, :86) -> sigma = 0
    coef = torch.stack((1 / abar ** 0.5, (1 - abar) ** 0.5, abar_next ** 0.5, (1 - abar_next - 0.0) ** 0.5, upd), dim=1)
    if cond_img is not None:
        cond_img = cond_img.to(dev)
    if labels_tensor is not None:
        labels_tensor = labels_tensor.to(dev)
    r = _StepRunner(diffusion_net, x_t, steps, coef, cond_img, labels_tensor, None, _use_graph(use_graph), cfg_scale)
    x0 = torch.empty_like(r.x)

    def body():
        r.begin()
        ops.ddim_update(r.x, r.eps(), r.coef_cur, x0_out=x0, cursor=r.cursor, nsteps=r.nsteps)

    def on_step(i):
        printProgressBar(iteration=max_noise - steps[i], total=max_noise - min_noise, prefix='Iterations:',
                         suffix='Complete', length=50, log=log)

    with torch.no_grad():
        r.run(body, on_step)
    return x0 if steps[-1] == 1 else r.x


def cold_diffusion_sampling(diffusion_net, noise_degradation, x_t, noise, min_noise=1, max_noise=1_000, cond_img=None,
                            labels_tensor=None, skip_step_size=10, device="cpu", log=print, *, use_graph=None):
    """x0-prediction sampler of the SR path (reference :150-217): x_t <- x_t - D(x0_hat, t) + D(x0_hat, t-1) with
    D(x0, t) = sqrt(abar_t) x0 + sqrt(1-abar_t) noise; both degradations and the update are one fused kernel and the
    step is replayed as a hipGraph like the other samplers.  Returns the last x0 estimate."""
    _require_cuda(x_t)
    dev = x_t.device
    diffusion_net.eval()
    steps = list(range(max_noise, min_noise - 1, -skip_step_size))
    if min_noise not in steps:
        steps = steps + [min_noise]
    abar = noise_degradation.get_timestep_params(torch.tensor(steps))[2].cpu()
    abar_next = torch.cat((abar[1:], abar[-1:]))
    upd = torch.ones_like(abar)
    upd[-1] = 0.0                                                    # no update after the last reconstruction
    coef = torch.stack((abar ** 0.5, (1 - abar) ** 0.5, abar_next ** 0.5, (1 - abar_next) ** 0.5, upd), dim=1)
    if cond_img is not None:
        cond_img = cond_img.to(dev).float().contiguous()
    if labels_tensor is not None:
        labels_tensor = labels_tensor.to(dev)
    noise = noise.to(dev).float().contiguous()
    r = _StepRunner(diffusion_net, x_t, steps, coef, cond_img, labels_tensor, None, _use_graph(use_graph))
    x0_hat = torch.empty((x_t.shape[0], diffusion_net.out_channel) + tuple(x_t.shape[2:]), dtype=torch.float32, device=dev)

    def body():
        r.begin()
        ops.cold_update(r.x, r.eps(), noise, r.coef_cur, x0_keep=x0_hat, cursor=r.cursor, nsteps=r.nsteps)   # the net predicts x0

    def on_step(i):
        if i < len(steps) - 1:
            printProgressBar(iteration=max_noise - steps[i], total=max_noise - min_noise, prefix='Iterations:',
                             suffix='Complete', length=50, log=log)

    with torch.no_grad():
        r.run(body, on_step)
    return x0_hat
