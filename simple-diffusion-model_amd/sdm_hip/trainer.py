"""Training-step engine for noise-prediction diffusion training on MI355X.

Mirrors the body of the reference loop (train_diffusion.py:307-366): eps ~ N(0,I), t ~ U{min..max-1},
x_t = q(x_t|x_0), eps_hat = U_Net(x_t, t, labels), loss = MSE, backward, Adam(lr, betas=(0.5, 0.999)).
What is new relative to the reference:
  * gradients live in ONE flat fp32 buffer laid out in backward-completion order, each parameter's
    gradient being a strided view with the parameter's own (tap-major) layout;
  * data parallelism: one process per GPU, the flat buffer is all-reduced in buckets over RCCL/xGMI on
    a side stream as soon as the blocks that write a bucket have finished their backward, and the fused
    Adam for a bucket starts when its all-reduce has landed (overlap with the rest of backward);
  * Adam is one fused multi-tensor kernel (sdm_adam_multi) with torch.optim.Adam's state layout.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import lib, ops
from .dp import BucketedAllReduce, make_exchange


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0) semantics on flat buffers; state_dict() is
    interchangeable with torch.optim.Adam's (state only for live parameters, keyed by their index in
    ``net.parameters()``), so reference checkpoints' "optimizer" entries round-trip."""

    def __init__(self, net, params_grads, lr, betas=(0.5, 0.999), eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, lr, betas, eps
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)]
        self.step_count = 0
        self.items = params_grads                     # [(param, grad)] in flat order
        dev = params_grads[0][0].device
        self.m = [torch.zeros_like(g) for _, g in params_grads]
        self.v = [torch.zeros_like(g) for _, g in params_grads]
        jobs = (lib.AdamJob * len(params_grads))()
        self.max_n = 0
        for j, ((p, g), m, v) in enumerate(zip(params_grads, self.m, self.v)):
            assert p.stride() == g.stride() == m.stride(), "Adam state must share the parameter layout"
            jobs[j].p, jobs[j].g, jobs[j].m, jobs[j].v, jobs[j].n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
            self.max_n = max(self.max_n, p.numel())
        self.jobs_host = jobs
        self.jobs_dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.njobs = len(params_grads)
        self.job_bytes = C.sizeof(lib.AdamJob)
        self.token = object()                         # this optimizer's identity for ops.weights_written
        self.adopted = 0

    def adopt_bf16_copies(self):
        """bf16 training path: the contractions read bf16 copies of the weights (and transposed copies for the data
        gradients), cached per module and refreshed by a cast launch whenever the fp32 master changed -- ~470 launches and
        ~1.1 ms of a ~25 ms step.  After the first step those copies exist; from here on sdm_adam_multi writes them together
        with the master (their addresses go into the job table), and custom_layers.bf16_weight stops re-casting them.
        Call on the stream the next forward runs on, after the optimizer stream has been joined.  Returns the number of
        copies adopted."""
        from models.custom_layers import bf16_weight
        index = {id(p): j for j, (p, _) in enumerate(self.items)}
        n = 0
        for mod in self.net.modules():
            cache = mod.__dict__.get("_bf16_cache")
            w = getattr(mod, "_parameters", {}).get("weight")
            if not cache or w is None or id(w) not in index:
                continue
            job = self.jobs_host[index[id(w)]]
            for key in list(cache):
                transposed = key in ("taps_t", "wt")
                if key not in ("taps", "w", "taps_t", "wt"):
                    continue
                # bring the copy in step with the master now (it was cast before the last optimizer step)
                cache[key] = (("stale",), cache[key][1])
                src = (lambda m=mod: m.taps()) if key.startswith("taps") else \
                      ((lambda m=mod: m.weight.detach()[None]) if transposed else (lambda m=mod: m.weight))
                buf = bf16_weight(mod, key, src, transposed=transposed, pad_to=8 if key == "taps_t" else 1)
                if transposed:
                    T, Cc, ldd = buf.shape
                    R = w.numel() // (T * Cc)
                    job.w16t, job.R, job.Cc, job.ldd = buf.data_ptr(), R, Cc, ldd
                else:
                    assert buf.numel() == w.numel()
                    job.w16 = buf.data_ptr()
                n += 1
        if n:
            self.jobs_dev.copy_(torch.frombuffer(bytearray(bytes(self.jobs_host)), dtype=torch.uint8))
            ops.weights_written(self.token)           # this optimizer is the writer the copies follow from here on
            for mod in self.net.modules():
                cache = mod.__dict__.get("_bf16_cache")
                w = getattr(mod, "_parameters", {}).get("weight")
                if not cache or w is None or id(w) not in index:
                    continue
                for key in list(cache):
                    if key in ("taps", "w", "taps_t", "wt"):
                        cache[key] = ((w.data_ptr(), w._version, None), cache[key][1], (self.token, ops.WRITER_SWITCHES[0]))
        self.adopted = n
        return n

    def step(self, first=0, count=None, grad_scale=1.0, advance=True, hyper=None):
        """Update jobs [first, first+count).  ``advance`` bumps the step counter (once per optimiser step).
        ``hyper`` (device [lr, step]) makes lr / bias corrections device-resident (graph replay)."""
        if advance:
            self.step_count += 1
        ops.weights_written(self.token)               # weights change behind torch's version counters
        count = self.njobs - first if count is None else count
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** max(self.step_count, 1)
        bc2 = 1.0 - b2 ** max(self.step_count, 1)
        max_n = max(self.items[j][0].numel() for j in range(first, first + count))
        if hyper is not None:
            bc1 = bc2 = 1.0
        lib.call("sdm_adam_multi", self.jobs_dev.data_ptr() + first * self.job_bytes, count, max_n, lr, b1, b2, self.eps,
                 bc1, bc2, grad_scale, None if hyper is None else hyper.data_ptr(), ops._stream())

    def zero_grad(self, set_to_none=True):
        pass                                          # gradients are overwritten by every backward

    def state_dict(self):
        index = {id(p): i for i, p in enumerate(self.net.parameters())}
        state = {}
        for (p, _), m, v in zip(self.items, self.m, self.v):
            if self.step_count:
                state[index[id(p)]] = dict(step=torch.tensor(float(self.step_count)), exp_avg=m, exp_avg_sq=v)
        group = dict(self.param_groups[0])
        group.update(maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(index))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        index = {id(p): i for i, p in enumerate(self.net.parameters())}
        for (p, _), m, v in zip(self.items, self.m, self.v):
            st = sd["state"].get(index[id(p)])
            if st is not None:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                self.step_count = int(st["step"])
        self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]


class DiffusionTrainer:
    def __init__(self, net, schedule, lr=2e-5, betas=(0.5, 0.999), min_noise_step=1, max_actual_noise_step=1000,
                 bucket_mb=64, process_group=None, seed=None, use_graph=None, sr=None, train_dtype=None, objective="eps",
                 dp_backend=None, dp_wire=None):
        """``sr=(lr_dim, sr_dim, cond_t)`` switches to the super-resolution objective of train_SR_diffusion.py:318-374:
        the net sees cat(q(x0, t), q(lowres(x0), cond_t)) and regresses x0 - lowres(x0).
        ``train_dtype=torch.bfloat16`` runs forward/backward in bf16 storage (U_Net.set_training_dtype); weights,
        gradients and Adam stay fp32.
        ``objective``: "eps" regresses the noise (train_diffusion.py:343-352, train_doodle_diffusion.py:300-312); "x0"
        regresses the clean image (cold diffusion, train_noise_cold_diffusion.py:335-342).  An image condition (doodle:
        ``cat((x_t, lbl_data), 1)``, train_doodle_diffusion.py:304) is passed per step as ``cond_img``.
        ``dp_backend`` ("torch" | "sdm") / ``dp_wire`` ("f32" | "bf16"): who runs the gradient exchange and in which wire
        format (sdm_hip/dp.py; defaults from SDM_DP_BACKEND / SDM_DP_WIRE, else torch.distributed in fp32)."""
        if objective not in ("eps", "x0"):
            raise ValueError("objective must be 'eps' or 'x0'")
        self.objective = objective
        self.sr = sr
        if train_dtype is not None:
            net.set_training_dtype(train_dtype)
        self.net, self.schedule = net, schedule
        # bf16 training: the fused Adam also writes the bf16 (and transposed bf16) weight copies (SDM_ADAM_BF16_COPIES=0: the
        # copies are re-cast from the fp32 masters every step, as before -- A/B)
        import os as _os
        self.adopt_bf16 = train_dtype is not None and _os.environ.get("SDM_ADAM_BF16_COPIES", "1") != "0"
        self.min_t, self.max_t = min_noise_step, max_actual_noise_step
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        dev = next(net.parameters()).device
        self.dev = dev
        order = net.grad_blocks()
        self.flat, self.grads, ranges = net.alloc_grads()
        self.items, self.block_range = [], {}
        for name, ps in order:
            j0 = len(self.items)
            self.items += [(p, self.grads[p]) for p in ps]
            self.block_range[name] = (ranges[name][0], ranges[name][1], j0, len(self.items))
        self.order = [n for n, _ in order]
        self.exchange = make_exchange(process_group, dp_backend, dp_wire, dev)
        self.reducer = BucketedAllReduce(self.flat, [(n, ranges[n][0], ranges[n][1]) for n in self.order], bucket_mb,
                                         process_group, exchange=self.exchange)
        # optimizer jobs covered by each bucket
        self.bucket_jobs = []
        for (s, e, first, last) in self.reducer.buckets:
            self.bucket_jobs.append((self.block_range[first][2], self.block_range[last][3]))
        self.opt = FusedAdam(net, self.items, lr, betas)
        self.opt_stream = torch.cuda.Stream(device=dev)
        # Optional (SDM_TRAIN_GRAPH=1, single GPU): capture the whole step (q-sample .. Adam) as one hipGraph.  Off by
        # default: on ROCm 7.2 replaying the ~500-node multi-stream graph measured SLOWER than eager launching
        # (10.1 vs 9.3 ms/step, MI355X, batch 4 64x64) -- the eager step is GPU-bound (host enqueue 7.5 ms/step).
        import os
        if use_graph is None:
            use_graph = os.environ.get("SDM_TRAIN_GRAPH", "0") == "1"
        self.use_graph = bool(use_graph) and self.world == 1
        self.graph, self.g_key, self.hyper = None, None, None
        # per-rank RNG stream for eps / t (and the CLI's flip mask): under data parallelism every rank must draw its own
        # noise and timesteps for its own images, so the seed is offset by the rank; unseeded runs take a fresh random seed
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.gen = torch.Generator(device=dev)
        if seed is not None:
            self.gen.manual_seed(int(seed) * self.world + self.rank)
        else:
            self.gen.seed()
        if self.world > 1:                            # identical replicas: rank 0's parameters everywhere
            for p in net.parameters():      # dense storage view: conv weights are permuted (non-contiguous) views
                dist.broadcast(p.data.as_strided((p.numel(),), (1,), p.storage_offset()), src=0, group=self.pg)

    # -- pieces -------------------------------------------------------------------------------
    def draw(self, x0):
        n = x0.shape[0]
        eps = torch.randn(x0.shape, device=self.dev, generator=self.gen)
        t = torch.randint(self.min_t, self.max_t, (n,), device=self.dev, generator=self.gen)
        return eps, t

    def _bucket_closed(self, name):
        """Backward hook: a top-level block's gradients have all been ENQUEUED (data gradients on the compute
        stream, weight gradients on the side stream).  If that closes a bucket, the optimizer stream waits for
        both streams, runs the bucket's all-reduce (DP) and then its fused Adam -- the compute stream never waits,
        so exchange and update overlap the rest of the backward."""
        i = self.reducer.closing.get(name)
        if i is None:
            return
        self.opt_stream.wait_stream(torch.cuda.current_stream(self.dev))
        if ops.SIDE.enabled:
            self.opt_stream.wait_stream(ops.SIDE.stream(self.dev))
        with torch.cuda.stream(self.opt_stream):
            self.reducer.block_done(name)                               # async all-reduce when world > 1
            _, work = self.reducer.pending[-1]
            if work is not None:
                work.wait()                                             # blocks the optimizer stream only
            j0, j1 = self.bucket_jobs[i]
            self.opt.step(j0, j1 - j0, grad_scale=1.0 / self.world, advance=False, hyper=getattr(self, "_hyper", None))
            # the bucket's gradients are consumed: clear them here, on the optimizer stream, for the next backward
            # (which accumulates) instead of a 0.5 GB memset at the head of the next step on the compute stream
            s, e = self.reducer.buckets[i][0], self.reducer.buckets[i][1]
            self.flat[s:e].zero_()
            self._cleared += e - s

    def sr_inputs(self, x0, eps, t):
        """(network input, regression target) of the super-resolution objective (train_SR_diffusion.py:318-374): the net sees
        cat(q(x0, t), q(lowres(x0), cond_t)) -- the SAME eps in both -- and regresses x0 - lowres(x0)."""
        lr_dim, sr_dim, cond_t = self.sr
        low = ops.area_resize(ops.area_resize(x0, (lr_dim, lr_dim)), (sr_dim, sr_dim))
        abar_c = self.schedule.alpha_bar_of(torch.full((x0.shape[0],), int(cond_t), dtype=torch.int64, device=self.dev))
        x_t = ops.q_sample(x0, eps, self.schedule.alpha_bar_of(t))
        return torch.cat((x_t, ops.q_sample(low, eps, abar_c)), dim=1), x0 - low

    def _body(self, x0, labels, eps, t, hyper=None, cond_img=None):
        if self.sr is not None:
            x_t, target = self.sr_inputs(x0, eps, t)
        else:
            x_t = ops.q_sample(x0, eps, self.schedule.alpha_bar_of(t))
            target = eps if self.objective == "eps" else x0
        if cond_img is not None:
            x_t = torch.cat((x_t, cond_img.to(torch.float32)), dim=1)
        if getattr(self, "_cleared", 0) != self.flat.numel():
            self.flat.zero_()                           # first step, or the previous one did not finish: start clean
        self._cleared = 0                               # backward accumulates (split-K atomics, column sums)
        self.reducer.begin_step()
        self._hyper = hyper
        self.net._block_done_hook = self._bucket_closed
        try:
            loss, _ = self.net.loss_and_grads(x_t, t, labels, target, grads=self.grads)
        finally:
            self.net._block_done_hook = None
        self.reducer.pending = []
        torch.cuda.current_stream(self.dev).wait_stream(self.opt_stream)   # next forward sees the updated weights
        if self.adopt_bf16 and not self.opt.adopted and not torch.cuda.is_current_stream_capturing():
            self.opt.adopt_bf16_copies()               # once, after the first step: Adam keeps the bf16 weight copies itself
        return loss

    def step(self, x0, labels=None, eps=None, t=None, cond_img=None):
        """One optimiser step on the local mini-batch; returns the loss tensor ([1], device)."""
        if eps is None or t is None:
            eps, t = self.draw(x0)
        x0 = x0.contiguous()
        self.opt.step_count += 1
        if not self.use_graph or cond_img is not None:
            return self._body(x0, labels, eps, t, cond_img=cond_img)
        key = (tuple(x0.shape), None if labels is None else tuple(labels.shape))
        if self.g_key != key:
            if self.hyper is None:
                self.hyper = torch.zeros(2, dtype=torch.float32, device=self.dev)
            self.g_in = [x0.clone(), None if labels is None else labels.float().contiguous().clone(), eps.clone(), t.clone()]
            self.hyper[0] = self.opt.param_groups[0]["lr"]
            self.hyper[1] = float(self.opt.step_count)
            self._lr_dev = self.opt.param_groups[0]["lr"]
            # warm-up outside capture (sizes scratch buffers, builds job tables, sets kernel attributes) on a throw-away
            # copy of the training state, so that one call still performs exactly one optimiser step
            backup = [(p.data.clone(), m.clone(), v.clone()) for (p, _), m, v in zip(self.opt.items, self.opt.m, self.opt.v)]
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._body(*self.g_in, hyper=self.hyper)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            for ((p, _), m, v), (bp, bm, bv) in zip(zip(self.opt.items, self.opt.m, self.opt.v), backup):
                p.data.copy_(bp)
                m.copy_(bm)
                v.copy_(bv)
            del backup
            if self.adopt_bf16 and self.opt.adopted:
                self.opt.adopt_bf16_copies()       # the copies followed the warm-up step: bring them back to the restored weights
            self.graph = torch.cuda.CUDAGraph()
            with ops.capture_scope() as kept, torch.cuda.graph(self.graph):
                self.g_loss = self._body(*self.g_in, hyper=self.hyper)
                self.hyper[1:2].add_(1.0)
            self._graph_kept = kept       # library buffers the graph points at live exactly as long as the graph
            self.g_key = key
            return self._replay(x0, labels, eps, t)
        return self._replay(x0, labels, eps, t)

    def _replay(self, x0, labels, eps, t):
        gi = self.g_in
        gi[0].copy_(x0)
        if labels is not None:
            gi[1].copy_(labels)
        gi[2].copy_(eps)
        gi[3].copy_(t)
        lr = self.opt.param_groups[0]["lr"]
        if lr != self._lr_dev:
            self.hyper[0:1].fill_(lr)
            self._lr_dev = lr
        self.graph.replay()
        ops.weights_written(self.opt.token)   # the replayed Adam rewrote the weights: bf16 copies of no-grad forwards are stale
        return self.g_loss
