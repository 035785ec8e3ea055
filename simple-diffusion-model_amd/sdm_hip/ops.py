"""Host-side launch helpers: torch tensors in, C-ABI calls out.  No arithmetic happens in
Python; torch is used for device memory and streams only.

Activations are NHWC "pixel rows x channels" (class Act); conv weights are tap-major
``[taps, c_out, c_in]`` views of the parameters (see models/custom_layers.py).
"""
import ctypes as C
import os

import torch

from . import lib
from .lib import ACT_NONE, ACT_SILU, ACT_TANH, IgemmDesc, Rows, WgradDesc


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


_override = [None]      # raw stream that launches go to inside a _SideStream.fork() block


def _stream():
    # torch's current stream on the current device as a raw hipStream_t (the C accessors: the Python wrapper
    # torch.cuda.current_stream() costs ~7 us, which at ~300 launches per step was a quarter of the host time)
    o = _override[0]
    return o if o is not None else _raw_stream(_cur_device())


class _Fork:
    """`with` block whose launches go to the given raw stream (no torch stream switch: nothing inside allocates)."""
    __slots__ = ("raw",)

    def __init__(self, raw):
        self.raw = raw

    def __enter__(self):
        _override[0] = self.raw

    def __exit__(self, *exc):
        _override[0] = None
        return False


def _p(t):
    return None if t is None else t.data_ptr()


class Act:
    """An activation as pixel rows: ``t`` is [n*h*w, c] (row pitch t.stride(0)); ``t2`` an optional second
    tensor concatenated on channels; ``scale``/``shift`` an optional pending per-(image,channel) affine
    (deferred GroupNorm+AdaGN).  ``z`` is the pre-activation kept for backward when training."""
    __slots__ = ("t", "n", "h", "w", "t2", "scale", "shift", "z", "stats")

    def __init__(self, t, n, h, w, t2=None, scale=None, shift=None, z=None):
        self.t, self.n, self.h, self.w, self.t2 = t, n, h, w, t2
        self.scale, self.shift, self.z = scale, shift, z
        self.stats = None       # [n, c, 2] {sum, sum of squares} per (image, channel) when the producing conv emitted them

    @property
    def c1(self):
        return self.t.shape[1]

    @property
    def c(self):
        return self.t.shape[1] + (self.t2.shape[1] if self.t2 is not None else 0)

    @property
    def rows(self):
        return self.n * self.h * self.w

    def with_affine(self, scale, shift):
        return Act(self.t, self.n, self.h, self.w, self.t2, scale, shift, self.z)


def fill_rows(r, a, h=None, w=None):
    r.p0 = _p(a.t)
    r.p1 = _p(a.t2)
    r.csplit = a.c1
    r.ld0 = a.t.stride(0)
    r.ld1 = a.t2.stride(0) if a.t2 is not None else 0
    r.scale = _p(a.scale)
    r.shift = _p(a.shift)
    r.aff_nstride = 0 if (a.scale is None or a.scale.shape[0] == 1) else a.scale.stride(0)
    r.H = a.h if h is None else h
    r.W = a.w if w is None else w


def plain_rows(r, t, rows=None):
    r.p0 = _p(t)
    r.p1 = None
    r.csplit = 1 << 30
    r.ld0 = t.stride(0)
    r.ld1 = 0
    r.scale = r.shift = None
    r.aff_nstride = 0
    r.H = t.shape[0] if rows is None else rows
    r.W = 1


class _SideStream:
    """Second HIP stream for weight-gradient kernels.  A conv's weight gradient and data gradient are independent,
    and at small batch neither fills 256 CUs on its own, so they run concurrently: fork() makes the side stream
    wait for everything enqueued so far on the current stream, join() makes the current stream wait for the side
    stream.  Tensors handed to fork() are registered with the caching allocator (record_stream) so their memory is
    not reused while the side stream still reads them.  SDM_WGRAD_STREAM=0 disables it."""

    def __init__(self):
        import os
        self.enabled = os.environ.get("SDM_WGRAD_STREAM", "1") != "0"
        self.streams = {}
        self.keep = []
        self.dirty = False

    def stream(self, device):
        st = self.streams.get(device)
        if st is None:
            st = self.streams[device] = self._masked_stream(device) or torch.cuda.Stream(device=device)
        return st

    @staticmethod
    def _masked_stream(device):
        """SDM_WGRAD_CUS=n (experiment switch, default off): the side stream's queue may use only the first n bits of the CU mask
        (hipExtStreamCreateWithCUMask), so the weight gradients leave the other CUs to the forward / data-gradient chain.
        Measured on the headline and the C4 bf16 step: profiles/r04_wgrad_cumask.txt."""
        import os
        n = int(os.environ.get("SDM_WGRAD_CUS", "0") or 0)
        if n <= 0:
            return None
        hip = C.CDLL("libamdhip64.so")
        words = (C.c_uint32 * 8)(*[(0xFFFFFFFF if n >= 32 * (i + 1) else ((1 << max(0, n - 32 * i)) - 1)) for i in range(8)])
        raw = C.c_void_p()
        with torch.cuda.device(device):
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(raw), 8, words)
        if rc != 0 or not raw.value:
            raise lib.SdmError(f"SDM_WGRAD_CUS={n}: hipExtStreamCreateWithCUMask failed (hip error {rc})")
        return torch.cuda.ExternalStream(raw.value, device=device)

    def fork(self, *tensors):
        """Tensors read on the side stream are kept referenced until join(): their memory cannot be handed out again
        (on the compute stream) before the side stream is done with them -- also valid under hipGraph capture."""
        dev = tensors[0].device
        side = self.stream(dev).cuda_stream
        lib.call("sdm_stream_order", _raw_stream(dev.index), side)       # side waits for what main has enqueued so far
        self.keep.extend(t for t in tensors if t is not None)
        self.dirty = True
        return _Fork(side)

    def join(self):
        if self.dirty:
            for dev, side in self.streams.items():
                lib.call("sdm_stream_order", side.cuda_stream, _raw_stream(dev.index))
            self.dirty = False
        self.keep.clear()


SIDE = _SideStream()


class _StreamKeyed:
    """Library-owned device state that ONE stream may use at a time (zero-filled scratch, the cooperative GroupNorm's
    exchange words): one instance per (device, stream that launches go to) outside graph capture, one per capture_scope
    inside -- so two forwards running concurrently on two streams of a process, or a replaying graph beside eager
    launches, never meet in the same words.  A capture's instances are made when the scope is entered (allocation is not
    possible once the capture runs), sized like the largest instance the warm-up run left on the device."""
    _all = []

    def __init__(self):
        self.by_stream = {}
        _StreamKeyed._all.append(self)

    def _prealloc(self, device_index):
        """Tensor for a capture that is about to start on ``device_index`` (None: nothing to pre-size from)."""
        raise NotImplementedError

    def _slot(self, device):
        scope = capture_scope.current()
        if scope is not None:
            return scope.local, (id(self), device.index)
        return self.by_stream, (device.index, _stream())


class capture_scope:
    """Lifetime of the library-owned buffers a captured hipGraph points at.

    A captured graph bakes in the ADDRESSES of buffers that live outside its own memory pool: the zero-filled scratch
    (_Workspace) and the y_scale job tables / output buffers (LinearBank).  Those caches are rebuilt when a shape key
    changes; if the old set were simply dropped, a later replay would read freed -- and soon reused -- memory (the 06:18
    fault of round 1: the training-step graph replayed after a checkpoint-plot forward with another embedding batch had
    rebuilt the y_scale tables, and dereferenced a job table that the allocator had handed to someone else).  So whoever
    captures a graph opens a scope; every cache that hands out a buffer during the capture registers it with keep(); the
    scope's list is stored next to the graph and dies with it.  The caches themselves can then be bounded (LRU) without
    ever freeing what a live graph still needs.

    The scope also OWNS the single-stream state of the capture (_StreamKeyed): the graph's launches use buffers of their
    own, made here, before the capture starts."""
    _stack = []

    @staticmethod
    def current():
        return capture_scope._stack[-1] if capture_scope._stack else None

    _serial = [0]

    def __enter__(self):
        self.kept = []
        self.local = {}
        capture_scope._serial[0] += 1
        self.serial = capture_scope._serial[0]
        if torch.cuda.is_available():
            dev = torch.cuda.current_device()
            for owner in _StreamKeyed._all:
                t = owner._prealloc(dev)
                if t is not None:
                    self.local[(id(owner), dev)] = t
                    self.kept.append(t)
            for bank in list(LinearBank._all or ()):
                t = bank.prealloc_for_capture(self)
                if t is not None:
                    self.kept.append(t)
        capture_scope._stack.append(self)
        return self.kept

    def __exit__(self, *exc):
        capture_scope._stack.pop()
        return False


def keep(obj):
    """Register ``obj`` with the graph capture in progress (no-op outside a capture_scope)."""
    if capture_scope._stack and obj is not None:
        capture_scope._stack[-1].kept.append(obj)


def _all_zero(owner):
    """True when every buffer of a _StreamKeyed owner is zero (tests of the zero-filled contract)."""
    return all(float(b.abs().max()) == 0.0 for b in owner.by_stream.values() if b.numel())


class _Workspace(_StreamKeyed):
    """Zero-filled scratch (split-K partial sums, GroupNorm-backward channel sums), grown on demand; the kernels leave
    what they touched zero-filled again.  One buffer per launching stream (_StreamKeyed).  An outgrown buffer is dropped: a
    captured hipGraph that still points at it holds its own reference (capture_scope)."""

    def _prealloc(self, device_index):
        n = max((b.numel() for (d, _), b in self.by_stream.items() if d == device_index), default=0)
        return torch.zeros(n, dtype=torch.float32, device=torch.device("cuda", device_index)) if n else None

    def get(self, nbytes, device):
        if nbytes == 0:
            return None
        store, key = self._slot(device)
        buf = store.get(key)
        if buf is None or buf.numel() * 4 < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise lib.SdmError("zero-filled scratch (split-K / GroupNorm backward) must be sized by a warm-up call before "
                                   "graph capture")
            buf = store[key] = torch.zeros((nbytes + 3) // 4 + 1024, dtype=torch.float32, device=device)   # zero-filled contract
        keep(buf)
        return buf


WS = _Workspace()
GNWS = _Workspace()     # scratch of sdm_gn_adagn_bwd (same zero-filled contract; separate so sizes do not interact)
GNSUMS = _Workspace()   # per-(image, channel) GroupNorm sums a bf16 conv epilogue emits; sdm_gn_finalize leaves them zero
CONV_GN_STATS = [True]  # bf16 rows: GroupNorm statistics in the producing conv's epilogue (False: separate pass)

# ---- tap tables -----------------------------------------------------------------------------
_C3 = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
# ConvTranspose2d(k4,s2,p1): output row 2y+a takes kernel rows {1,3} (a=0) or {0,2} (a=1)
_CT_1D = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}      # (k, d)
# data-gradient of Conv2d(k3,s2,p1): input row 2y+a receives from kernel rows {1} (a=0) or {0,2} (a=1)
_S2D_1D = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}


def _cols(taps):
    """(n, [dy...], [dx...], [w...]) -- column form of a static tap table, assigned to the descriptor by slices."""
    return (len(taps), [t[0] for t in taps], [t[1] for t in taps], [t[2] for t in taps])


def _set_taps(d, phase, cols):
    n, dys, dxs, ws = cols if isinstance(cols, tuple) else _cols(cols)
    d.ntaps[phase] = n
    d.tap_dy[phase][:n] = dys
    d.tap_dx[phase][:n] = dxs
    d.tap_w[phase][:n] = ws


_C3_COLS = _cols(_C3)
_C3_NEG_COLS = _cols([(-dy, -dx, wi) for (dy, dx, wi) in _C3])
_CT_PHASE_COLS = {(a, b): _cols([(dy, dx, ky * 4 + kx) for (ky, dy) in _CT_1D[a] for (kx, dx) in _CT_1D[b]])
                  for a in (0, 1) for b in (0, 1)}
_S2D_PHASE_COLS = {(a, b): _cols([(dy, dx, ky * 3 + kx) for (ky, dy) in _S2D_1D[a] for (kx, dx) in _S2D_1D[b]])
                   for a in (0, 1) for b in (0, 1)}
_CT_DGRAD_COLS = _cols([(ky - 1, kx - 1, ky * 4 + kx) for ky in range(4) for kx in range(4)])
_WG_CT = [(ky - 1, kx - 1, 0, 0, ky * 4 + kx) for ky in range(4) for kx in range(4)]
_WG_C3 = [(0, 0, ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
_WG_CT_COLS = tuple([t[i] for t in _WG_CT] for i in range(5))
_WG_C3_COLS = tuple([t[i] for t in _WG_C3] for i in range(5))


BF16 = torch.bfloat16
WEIGHT_EPOCH = [0]      # bumped whenever weights are rewritten behind torch's back (fused Adam): bf16 copies go stale
_LAST_WRITER = [None]   # token of the optimizer that bumped it last
WRITER_SWITCHES = [0]   # how often the identity of that writer changed


def weights_written(token):
    """An optimizer (identified by ``token``) has rewritten weights through raw pointers.  bf16 weight copies tagged with an
    older epoch are stale -- except copies that this very optimizer keeps up to date itself (FusedAdam.adopt_bf16_copies:
    sdm_adam_multi writes them together with the fp32 master), which stay valid for as long as no OTHER writer steps in."""
    WEIGHT_EPOCH[0] += 1
    if _LAST_WRITER[0] is not token:
        _LAST_WRITER[0] = token
        WRITER_SWITCHES[0] += 1


def _launch_igemm(d, device, bf16=False, out_f32=False):
    """bf16=True: operands (and the output unless out_f32) are bf16, see sdm_igemm_bf16 (forward / sampling path).
    Returns the library's code: lib.DEFERRED when d.defer_epilogue was set and raw split-K sums were left in WS."""
    need = lib.raw("sdm_igemm_bf16_workspace_bytes" if bf16 else "sdm_igemm_workspace_bytes")(C.byref(d))
    ws = WS.get(need, device)
    d.ws = _p(ws)
    d.ws_bytes = ws.numel() * 4 if ws is not None else 0
    _LAST_WS[0] = ws
    if bf16:
        return lib.call("sdm_igemm_bf16", C.byref(d), int(out_f32), _stream())
    return lib.call("sdm_igemm", C.byref(d), _stream())


_LAST_WS = [None]       # the scratch the last contraction launch was given (a Deferred output lives there)


class Deferred:
    """A conv output whose split-K finishing pass has NOT run: raw sums [rows, c] in the shared zero-filled scratch
    (the launching stream's WS buffer), plus what the finishing pass would have applied.  Consumed -- and the scratch cleared -- by the very next
    launch (gn_apply_fused); nothing else may touch WS in between."""
    __slots__ = ("ws", "n", "h", "w", "c", "bias", "act", "dtype", "out", "z")

    def __init__(self, ws, n, h, w, c, bias, act, dtype=torch.float32, out=None, z=None):
        self.ws, self.n, self.h, self.w, self.c, self.bias, self.act, self.dtype = ws, n, h, w, c, bias, act, dtype
        self.out, self.z = out, z        # training: the (still unwritten) rows the finishing kernel must produce


def to_bf16(t, out=None):
    """Round-to-nearest-even bf16 copy of an fp32 tensor, keeping its memory layout (weights).  ``out``: a copy made
    earlier, refreshed in place (its address stays valid for captured graphs)."""
    src = t.detach()
    dense = src.as_strided((src.numel(),), (1,)) if not src.is_contiguous() else src.reshape(-1)
    if out is None:
        flat = torch.empty(src.numel(), dtype=BF16, device=src.device)
        out = flat.as_strided(src.shape, src.stride())
    lib.call("sdm_cast_bf16", _p(dense), _p(out), src.numel(), _stream())
    return out


def to_bf16_t(w3, out=None, pad_to=1):
    """Transposed bf16 copy of a contiguous fp32 [T, R, C] tensor -> [T, C, Rp] (the weight layout a bf16 DATA gradient
    contracts against: tap-major conv weights [tap][c_out][c_in] -> [tap][c_in][c_out]; Linear [O][K] -> [1][K][O]).
    Rp = R rounded up to ``pad_to`` (zero-filled): the reduction granule of the bf16 contraction is 8."""
    T, R, C_ = w3.shape
    assert w3.is_contiguous()
    Rp = (R + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((T, C_, Rp), dtype=BF16, device=w3.device)
    lib.call("sdm_cast_bf16_t", _p(w3), _p(out), T, R, C_, Rp, _stream())
    return out


def nchw_to_rows_bf16(x, cp=8):
    """NCHW fp32 [n, c, h, w] -> NHWC bf16 rows [n*h*w, cp] (channels [c, cp) zero): the image side of the edge convs in
    the row format of the matrix-core kernels."""
    n, c, h, w = x.shape
    out = torch.empty((n * h * w, cp), dtype=BF16, device=x.device)
    lib.call("sdm_nchw_to_rows_bf16", _p(x), _p(out), n, c, h, w, cp, _stream())
    return out


def nchw_to_rows(x, cp=4):
    """NCHW fp32 [n, c, h, w] -> NHWC fp32 rows [n*h*w, cp] (channels [c, cp) zero): the image side of the edge convs as
    pixel rows for the matrix-core weight-gradient kernel."""
    n, c, h, w = x.shape
    out = torch.empty((n * h * w, cp), dtype=torch.float32, device=x.device)
    lib.call("sdm_nchw_to_rows", _p(x), _p(out), n, c, h, w, cp, _stream())
    return out


def _base_desc(M, N, K):
    d = IgemmDesc()
    d.M, d.N, d.K, d.batch, d.bdiv = M, N, K, 1, 1
    d.sstride = d.ostride = d.nphase = 1
    d.alpha = 1.0
    return d


def conv_out_hw(h, w, kind):
    if kind == "c3s1":
        return h, w
    if kind == "c3s2":
        return (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if kind == "ct4":
        return 2 * h, 2 * w
    raise ValueError(kind)


def conv_fwd(x, wt, bias, kind, act=ACT_SILU, save_z=False, defer=False, gn_stats=False):
    """x: Act; wt: [taps, c_out, c_in] contiguous view; returns Act (plain).  ``defer`` (fp32, no save_z): if the library
    splits K it skips its finishing pass and a Deferred is returned instead (see gn_apply_fused)."""
    T, cout, cin = wt.shape
    assert cin == x.c, (cin, x.c)
    ho, wo = conv_out_hw(x.h, x.w, kind)
    bf = wt.dtype == BF16
    if bf and (x.scale is not None or x.t.dtype != BF16):
        raise lib.SdmError("bf16 conv: needs plain bf16 rows (materialise a pending affine first)")
    out = torch.empty((x.n * ho * wo, cout), dtype=wt.dtype, device=wt.device)
    z = torch.empty_like(out) if save_z else None
    if kind == "ct4":
        d = _base_desc(x.rows, cout, cin)
        d.Ht, d.Wt = x.h, x.w
        d.nphase, d.ostride = 4, 2
        for a in (0, 1):
            for b in (0, 1):
                ph = a * 2 + b
                _set_taps(d, ph, _CT_PHASE_COLS[(a, b)])
                d.ph_oy[ph], d.ph_ox[ph] = a, b
    else:
        s = 1 if kind == "c3s1" else 2
        d = _base_desc(x.n * ho * wo, cout, cin)
        d.Ht, d.Wt, d.sstride = ho, wo, s
        _set_taps(d, 0, _C3_COLS)
    fill_rows(d.a, x)
    d.b, d.ldb, d.b_tap_stride, d.b_kmajor = _p(wt), cin, cout * cin, 0
    d.Ho, d.Wo = ho, wo
    d.out, d.ldo = _p(out), out.stride(0)
    d.bias, d.act, d.out_preact = _p(bias), act, _p(z)
    d.defer_epilogue = int(bool(defer) and (not save_z or not bf))      # fp32: the fused training tail writes z as well
    sums = None
    if gn_stats and bf and CONV_GN_STATS[0] and kind == "c3s1":
        sums = GNSUMS.get(x.n * cout * 2 * 4, wt.device)       # zero-filled; consumed (and cleared) by sdm_gn_finalize
        d.gn_sums = _p(sums)
    rc = _launch_igemm(d, wt.device, bf16=bf)
    if rc == lib.DEFERRED:
        return Deferred(_LAST_WS[0], x.n, ho, wo, cout, bias, act, BF16 if bf else torch.float32, out=out if save_z else None, z=z)
    y = Act(out, x.n, ho, wo, z=z)
    if rc == lib.STATS_DONE:
        y.stats = sums
    return y


def conv_dgrad(g, g_hw, wt, kind, x_hw, n, res=None, res2=None, dact_z=None, dact_ncols=0, out=None):
    """Data gradient of conv_fwd.  g: [n*ho*wo, c_out] (pitch g.stride(0)); returns dx [n*h*w, c_in].
    fp32: ``wt`` is the forward weight [taps, c_out, c_in], read K-major.  bf16: ``wt`` is the TRANSPOSED copy
    [taps, c_in, c_out] (to_bf16_t), so that both operands stay reduction-contiguous; g / res / res2 / dact_z bf16."""
    bf = wt.dtype == BF16
    if bf:
        T, cin, cout = wt.shape
    else:
        T, cout, cin = wt.shape
    ho, wo = g_hw
    h, w = x_hw
    if out is None:
        out = torch.empty((n * h * w, cin), dtype=wt.dtype, device=wt.device)
    ga = Act(g, n, ho, wo)
    if kind == "c3s1":
        d = _base_desc(n * h * w, cin, cout)
        d.Ht, d.Wt = h, w
        _set_taps(d, 0, _C3_NEG_COLS)
    elif kind == "c3s2":
        ht, wt_ = (h + 1) // 2, (w + 1) // 2
        d = _base_desc(n * ht * wt_, cin, cout)
        d.Ht, d.Wt, d.nphase, d.ostride = ht, wt_, 4, 2
        for a in (0, 1):
            for b in (0, 1):
                ph = a * 2 + b
                _set_taps(d, ph, _S2D_PHASE_COLS[(a, b)])
                d.ph_oy[ph], d.ph_ox[ph] = a, b
    elif kind == "ct4":
        d = _base_desc(n * h * w, cin, cout)
        d.Ht, d.Wt, d.sstride = h, w, 2
        _set_taps(d, 0, _CT_DGRAD_COLS)
    else:
        raise ValueError(kind)
    fill_rows(d.a, ga)
    if bf:
        d.b, d.ldb, d.b_tap_stride, d.b_kmajor = _p(wt), cout, cout * cin, 0
    else:
        d.b, d.ldb, d.b_tap_stride, d.b_kmajor = _p(wt), cin, cout * cin, 1
    d.Ho, d.Wo = h, w
    d.out, d.ldo = _p(out), out.stride(0)
    d.act = ACT_NONE
    if res is not None:
        d.res, d.ldres = _p(res), res.stride(0)
    if res2 is not None:
        d.res2, d.ldres2 = _p(res2), res2.stride(0)
    if dact_z is not None:
        d.dact_z, d.ldz, d.dact_ncols = _p(dact_z), dact_z.stride(0), dact_ncols
    _launch_igemm(d, wt.device, bf16=bf)
    return out


def conv_wgrad(g, g_hw, x, kind, dwt, accumulate=2, dbias=None):
    """dwt[tap][co][ci] (+)= sum_p g[.][co] * x~[.][ci]; g: [n*ho*wo, c_out] rows; x: Act (input of the conv).
    dbias (zero-filled [c_out]) additionally receives the column sums of g (fused for plain-row gradients)."""
    T, cout, cin = dwt.shape
    ho, wo = g_hw
    d = WgradDesc()
    d.M, d.N, d.batch, d.bdiv = cout, cin, 1, 1
    d.images = x.n
    plain_rows(d.g, g)
    d.g.H, d.g.W = ho, wo
    fill_rows(d.x, x)
    if kind == "ct4":
        d.Hk, d.Wk = x.h, x.w
        d.g_stride, d.x_stride = 2, 1
        cols, nt = _WG_CT_COLS, 16
    else:
        s = 1 if kind == "c3s1" else 2
        d.Hk, d.Wk = ho, wo
        d.g_stride, d.x_stride = 1, s
        cols, nt = _WG_C3_COLS, 9
    d.ntaps = nt
    d.g_dy[:nt], d.g_dx[:nt], d.x_dy[:nt], d.x_dx[:nt], d.tap_w[:nt] = cols
    d.dw, d.ldw, d.w_tap_stride = _p(dwt), cin, cout * cin
    d.alpha, d.accumulate = 1.0, int(accumulate)
    if dbias is not None and kind != "ct4":
        d.dbias = _p(dbias)
    lib.call("sdm_wgrad_bf16" if g.dtype == BF16 else "sdm_wgrad", C.byref(d), _stream())
    if dbias is not None and kind == "ct4":        # gathered gradient rows: separate column sum
        colsum(g, dbias)


def linear_rows(x, W, b, act=ACT_NONE, res=None, save_z=False, out=None, ldres=None):
    """Token linear on the matrix cores: x [rows, K] (any pitch) @ W[O, K]^T + b (+ res).  ``ldres`` overrides the row
    pitch of ``res`` (0 = one row broadcast over all rows)."""
    rows, K = x.shape
    O = W.shape[0]
    bf = W.dtype == BF16
    if bf and (save_z or x.dtype != BF16 or (res is not None and res.dtype != BF16)):
        raise lib.SdmError("bf16 linear: forward-only path on bf16 rows")
    if out is None:
        out = torch.empty((rows, O), dtype=W.dtype, device=x.device)
    d = _base_desc(rows, O, K)
    d.Ht, d.Wt = rows, 1
    plain_rows(d.a, x)
    d.b, d.ldb, d.b_tap_stride, d.b_kmajor = _p(W), W.stride(0), 0, 0
    d.ntaps[0] = 1
    d.Ho, d.Wo = rows, 1
    d.out, d.ldo = _p(out), out.stride(0)
    d.bias, d.act = _p(b), act
    if res is not None:
        if ldres is None and res.shape[0] != rows:
            raise lib.SdmError(f"linear_rows: residual has {res.shape[0]} rows for {rows} output rows")
        d.res, d.ldres = _p(res), res.stride(0) if ldres is None else ldres
    _launch_igemm(d, x.device, bf16=bf)
    return out


def linear_rows_dgrad(g, W, res=None, out=None):
    """dx [rows, K] = g [rows, O] @ W[O, K] (+ res).  bf16: ``W`` is the transposed copy [K, O] (to_bf16_t)."""
    rows, O = g.shape
    bf = W.dtype == BF16
    K = W.shape[0] if bf else W.shape[1]
    if out is None:
        out = torch.empty((rows, K), dtype=W.dtype, device=g.device)
    d = _base_desc(rows, K, O)
    d.Ht, d.Wt = rows, 1
    plain_rows(d.a, g)
    d.b, d.ldb, d.b_tap_stride, d.b_kmajor = _p(W), W.stride(0), 0, 0 if bf else 1
    d.ntaps[0] = 1
    d.Ho, d.Wo = rows, 1
    d.out, d.ldo = _p(out), out.stride(0)
    if res is not None:
        d.res, d.ldres = _p(res), res.stride(0)
    _launch_igemm(d, g.device, bf16=bf)
    return out


def linear_rows_wgrad(g, x, dW, accumulate=2, dbias=None):
    """dW[O, K] (+)= g[rows, O]^T @ x[rows, K]  (+ dbias[O] += column sums of g); g / x fp32 or both bf16, dW fp32."""
    rows, O = g.shape
    K = x.shape[1]
    d = WgradDesc()
    d.M, d.N, d.batch, d.bdiv = O, K, 1, 1
    d.images, d.Hk, d.Wk = 1, rows, 1
    plain_rows(d.g, g)
    plain_rows(d.x, x)
    d.g_stride = d.x_stride = 1
    d.ntaps = 1
    d.dw, d.ldw, d.w_tap_stride = _p(dW), dW.stride(0), 0
    d.alpha, d.accumulate = 1.0, int(accumulate)
    d.dbias = _p(dbias)
    lib.call("sdm_wgrad_bf16" if g.dtype == BF16 else "sdm_wgrad", C.byref(d), _stream())


def _pad4(s):
    return (s + 3) // 4 * 4


TWO_PASS_ATTN = [os.environ.get("SDM_ATTN_TWOPASS", "1") != "0"]   # bf16 rows, S > 256: no S x S fp32 scores in HBM
TWO_PASS_MAX_DK = 512    # measured (tools/attn_probe.py): at head width 512 the second scores GEMM costs less than the fp32
                         # round trip + softmax pass it replaces (8 x 4096 tokens: 823 -> 697 us; 128 x 1024: 923 -> 854;
                         # 32 x 1024: 228 -> 214); at 1024 it costs more (16 x 1024: 166 -> 181 us)
NOGRAD_ATTN = [os.environ.get("SDM_NOGRAD_ATTN", "1") != "0"]   # short sequences, no-grad: probabilities never written
SMALL_ATTN = [os.environ.get("SDM_SMALL_ATTN", "1") != "0"]     # short sequences: the two-launch core of sdm_attn.hip
SMALL_ATTN_MAX_S = 256


SMALL_ATTN_MAX_WORK = 160 * 1024     # batch * heads * S^2 score elements


def _small_attn_ok(qkv, n, heads, S, dk, need_p=True):
    """The short-sequence cores are built for the launch-bound regime (sampling at batch 1-4, the 4x4 / 8x8 maps of a
    training step): one or a few workgroups per score tile, operands straight from L2.  Measured against the generic
    three-launch path (tools/scratch/attn_small_ab.py, profiles/r03_attn_small_ab.txt): they win while batch * heads * S^2
    stays under ~160k score elements (S = 256: one or two images, 20 vs 36 us; S = 64: up to ~32 images, 13-25 vs 26-30 us;
    S = 16: any batch) and lose beyond it (S = 256 at 128 images: 1007 vs 229 us), and the form that keeps P (training)
    only pays at S <= 64."""
    if not (SMALL_ATTN[0] and S <= SMALL_ATTN_MAX_S and dk % 16 == 0 and qkv.stride(0) % 8 == 0):
        return False
    if n * heads * S * S > SMALL_ATTN_MAX_WORK:
        return False
    if need_p or not NOGRAD_ATTN[0]:       # the P-keeping form would run
        return S <= 64
    return True


def attention_core_small(qkv, n, S, heads, dk, Sp, need_p=True):
    """Attention core of a short sequence in two launches (scores + column softmax on chip; P.V): returns (res, P) in the
    rows' type with P laid out [n*heads, S, Sp] exactly as the generic path leaves it (the training backward keeps it).
    ``need_p=False`` (no-grad forwards): the tile-per-workgroup form that never writes the probabilities; returns (res, None)."""
    dev, bf = qkv.device, int(qkv.dtype == BF16)
    if not need_p and NOGRAD_ATTN[0]:
        res = torch.empty((n * S, heads * dk), dtype=qkv.dtype, device=dev)
        nb = lib.raw("sdm_attn_small_nograd_workspace_bytes")(n * heads, S)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        lib.call("sdm_attn_small_nograd", _p(qkv), qkv.stride(0), n * heads, S, heads, dk, float(dk) ** -0.5, _p(res), res.stride(0), bf,
                 _p(ws), nb, _stream())
        return res, None
    P = torch.empty((n * heads, S, Sp), dtype=qkv.dtype, device=dev)
    res = torch.empty((n * S, heads * dk), dtype=qkv.dtype, device=dev)
    st = _stream()
    lib.call("sdm_attn_scores_small", _p(qkv), qkv.stride(0), n * heads, S, heads, dk, float(dk) ** -0.5, _p(P), Sp, bf, st)
    lib.call("sdm_attn_pv_small", _p(P), Sp, _p(qkv), qkv.stride(0), n * heads, S, heads, dk, _p(res), res.stride(0), bf, st)
    return res, P


def attention_core_fwd(qkv, n, S, heads, dk, keep_p=True):
    """qkv: [n*S, heads*3*dk] with per-head [q|k|v] chunks (custom_layers.py:138-141).
    Returns (res [n*S, heads*dk], P [n*heads, S, Sp]) with P the column-softmaxed scores (None when ``keep_p`` is False and
    the short-sequence no-grad form ran)."""
    dev = qkv.device
    Sp = _pad4(S)
    ld = qkv.stride(0)
    if _small_attn_ok(qkv, n, heads, S, dk, need_p=keep_p):
        return attention_core_small(qkv, n, S, heads, dk, Sp, need_p=keep_p)
    P = torch.empty((n * heads, S, Sp), dtype=torch.float32, device=dev)
    d = _base_desc(S, S, dk)
    d.batch, d.bdiv = n * heads, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, qkv, rows=S)
    d.a_bs0, d.a_bs1 = S * ld, 3 * dk
    d.b, d.ldb, d.b_kmajor = qkv.data_ptr() + 4 * dk, ld, 0
    d.b_bs0, d.b_bs1 = S * ld, 3 * dk
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(P), Sp, heads * S * Sp, S * Sp
    d.alpha = float(dk) ** -0.5
    _launch_igemm(d, dev)
    lib.call("sdm_col_softmax", _p(P), n * heads, S, Sp, _stream())
    res = torch.empty((n * S, heads * dk), dtype=torch.float32, device=dev)
    d = _base_desc(S, dk, S)
    d.batch, d.bdiv = n * heads, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, P[0], rows=S)
    d.a.ld0 = Sp
    d.a_bs0, d.a_bs1 = heads * S * Sp, S * Sp
    d.b, d.ldb, d.b_kmajor = qkv.data_ptr() + 8 * dk, ld, 1
    d.b_bs0, d.b_bs1 = S * ld, 3 * dk
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(res), heads * dk, S * heads * dk, dk
    _launch_igemm(d, dev)
    return res, P


def _pad8(s):
    return (s + 7) // 8 * 8


def attention_core_fwd_bf16(qkv, n, S, heads, dk, keep_p=False):
    """bf16 form of attention_core_fwd: scores are produced in fp32, the column softmax writes bf16 probabilities, V is
    transposed once so that P.V contracts two token-contiguous operands.  Returns res bf16 (and P with keep_p)."""
    dev = qkv.device
    Sp = _pad8(S)
    ld = qkv.stride(0)
    if _small_attn_ok(qkv, n, heads, S, dk, need_p=keep_p):
        res, P = attention_core_small(qkv, n, S, heads, dk, Sp, need_p=keep_p)
        return (res, P) if keep_p else res
    B = n * heads

    def scores_desc():
        d = _base_desc(S, S, dk)
        d.batch, d.bdiv = B, heads
        d.Ht, d.Wt = S, 1
        plain_rows(d.a, qkv, rows=S)
        d.a_bs0, d.a_bs1 = S * ld, 3 * dk
        d.b, d.ldb, d.b_kmajor = qkv.data_ptr() + 2 * dk, ld, 0
        d.b_bs0, d.b_bs1 = S * ld, 3 * dk
        d.ntaps[0] = 1
        d.Ho, d.Wo = S, 1
        d.alpha = float(dk) ** -0.5
        return d

    P = torch.empty((B, S, Sp), dtype=BF16, device=dev)
    if TWO_PASS_ATTN[0] and S % 8 == 0 and dk <= TWO_PASS_MAX_DK:
        # Two passes of the SAME scores GEMM instead of an S x S fp32 round trip through HBM (custom_layers.py:144-150; the
        # softmax runs down the columns -- over the queries -- so its statistics belong to the key):
        #   pass 1: per M-tile column statistics {max, sum exp} from the accumulators, nothing stored; folded per key;
        #   pass 2: the epilogue writes exp(s - m_j) / l_j once, as the bf16 probabilities P.V (and the backward) read.
        d = scores_desc()
        d.out, d.ldo = _p(P), Sp                 # (any output: the query only plans the tiles)
        tile_m = lib.raw("sdm_igemm_bf16_tile_m")(C.byref(d))
        if tile_m <= 0:
            raise lib.SdmError("attention (two-pass): " + lib.last_error())
        nt = (S + tile_m - 1) // tile_m
        stats = torch.empty((B, nt, S, 2), dtype=torch.float32, device=dev)
        d.col_stats, d.col_stats_bs = _p(stats), nt * S * 2
        d.out, d.ldo = None, 0
        _launch_igemm(d, dev, bf16=True)
        neg_max = torch.empty((B, S), dtype=torch.float32, device=dev)
        inv_sum = torch.empty((B, S), dtype=torch.float32, device=dev)
        lib.call("sdm_attn_colstats_combine", _p(stats), B, nt, S, _p(neg_max), _p(inv_sum), _stream())
        d = scores_desc()
        d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(P), Sp, heads * S * Sp, S * Sp
        d.act, d.col_shift, d.col_mul, d.colvec_bs = lib.ACT_EXP, _p(neg_max), _p(inv_sum), S
        _launch_igemm(d, dev, bf16=True)
    else:
        scores = torch.empty((B, S, Sp), dtype=torch.float32, device=dev)
        d = scores_desc()
        d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(scores), Sp, heads * S * Sp, S * Sp
        _launch_igemm(d, dev, bf16=True, out_f32=True)
        lib.call("sdm_col_softmax_bf16", _p(scores), _p(P), B, S, Sp, Sp, _stream())
        del scores
    Vt = torch.empty((n * heads, dk, Sp), dtype=BF16, device=dev)
    lib.call("sdm_transpose_bf16", qkv.data_ptr() + 4 * dk, ld, S * ld, 3 * dk, heads, _p(Vt), Sp, dk * Sp, n * heads, S, dk,
             _stream())
    res = torch.empty((n * S, heads * dk), dtype=BF16, device=dev)
    d = _base_desc(S, dk, Sp)
    d.batch, d.bdiv = n * heads, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, P[0], rows=S)
    d.a.ld0 = Sp
    d.a_bs0, d.a_bs1 = heads * S * Sp, S * Sp
    d.b, d.ldb, d.b_kmajor = _p(Vt), Sp, 0
    d.b_bs0, d.b_bs1 = heads * dk * Sp, dk * Sp
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(res), heads * dk, S * heads * dk, dk
    _launch_igemm(d, dev, bf16=True)
    return (res, P) if keep_p else res


def attention_core_bwd_bf16(dres, qkv, P, n, S, heads, dk):
    """bf16 backward of the attention core: returns dqkv bf16 [n*S, heads*3*dk] given dres bf16 [n*S, heads*dk] and the
    saved qkv / P (bf16).  dV and dK are token-reduction (weight-gradient shaped) products -> sdm_wgrad_bf16 into an fp32
    buffer; dP and dQ are row products -> sdm_igemm_bf16 with fp32 output; one cast at the end."""
    dev = qkv.device
    Sp = P.shape[2]
    ld = qkv.stride(0)
    dq32 = torch.zeros((n * S, heads * 3 * dk), dtype=torch.float32, device=dev)     # k / v slots take split-K atomics
    ldq = dq32.stride(0)
    st = _stream()
    B = n * heads

    def token_product(g3, x2d, x_off_elems, x_bs, slot):
        """dq32[slot][j][dd] = sum_i g3[b][i][j] * x[b][i][dd]"""
        w = WgradDesc()
        w.M, w.N, w.batch, w.bdiv = S, dk, B, heads
        w.images, w.Hk, w.Wk = 1, S, 1
        plain_rows(w.g, g3[0], rows=S)
        w.g.ld0 = Sp
        w.g_bs0, w.g_bs1 = heads * S * Sp, S * Sp
        plain_rows(w.x, x2d, rows=S)
        w.x.p0 = x2d.data_ptr() + 2 * x_off_elems
        w.x_bs0, w.x_bs1 = S * x2d.stride(0), x_bs
        w.g_stride = w.x_stride = 1
        w.ntaps = 1
        w.dw, w.ldw = dq32.data_ptr() + 4 * slot * dk, ldq
        w.w_bs0, w.w_bs1 = S * ldq, 3 * dk
        w.alpha, w.accumulate = 1.0, 2
        lib.call("sdm_wgrad_bf16", C.byref(w), st)

    # dV[j][dd] = sum_i P[i][j] dres[i][dd]
    token_product(P, dres, 0, dk, 2)
    # dP[i][j] = sum_dd dres[i][dd] v[j][dd]   (fp32, ahead of the softmax backward)
    dP = torch.empty((B, S, Sp), dtype=torch.float32, device=dev)
    d = _base_desc(S, S, dk)
    d.batch, d.bdiv = B, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, dres, rows=S)
    d.a_bs0, d.a_bs1 = S * dres.stride(0), dk
    d.b, d.ldb, d.b_kmajor = qkv.data_ptr() + 4 * dk, ld, 0
    d.b_bs0, d.b_bs1 = S * ld, 3 * dk
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(dP), Sp, heads * S * Sp, S * Sp
    _launch_igemm(d, dev, bf16=True, out_f32=True)
    dS = torch.empty((B, S, Sp), dtype=BF16, device=dev)
    lib.call("sdm_col_softmax_bwd_bf16", _p(P), Sp, _p(dP), Sp, _p(dS), Sp, B, S, float(dk) ** -0.5, st)
    del dP
    # dQ[i][dd] = sum_j dS[i][j] k[j][dd]: K transposed once so that both operands are contiguous along j
    Kt = torch.empty((B, dk, Sp), dtype=BF16, device=dev)
    lib.call("sdm_transpose_bf16", qkv.data_ptr() + 2 * dk, ld, S * ld, 3 * dk, heads, _p(Kt), Sp, dk * Sp, B, S, dk, st)
    d = _base_desc(S, dk, Sp)
    d.batch, d.bdiv = B, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, dS[0], rows=S)
    d.a.ld0 = Sp
    d.a_bs0, d.a_bs1 = heads * S * Sp, S * Sp
    d.b, d.ldb, d.b_kmajor = _p(Kt), Sp, 0
    d.b_bs0, d.b_bs1 = heads * dk * Sp, dk * Sp
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(dq32), ldq, S * ldq, 3 * dk
    _launch_igemm(d, dev, bf16=True, out_f32=True)
    # dK[j][dd] = sum_i dS[i][j] q[i][dd]
    token_product(dS, qkv, 0, 3 * dk, 1)
    dqkv = torch.empty((n * S, heads * 3 * dk), dtype=BF16, device=dev)
    lib.call("sdm_cast_bf16", _p(dq32), _p(dqkv), dq32.numel(), st)
    return dqkv


def attention_core_bwd(dres, qkv, P, n, S, heads, dk):
    """Returns dqkv [n*S, heads*3*dk] given dres [n*S, heads*dk]."""
    dev = qkv.device
    Sp = P.shape[2]
    ld = qkv.stride(0)
    hd = heads * dk
    dqkv = torch.zeros_like(qkv)          # the k / v slots are accumulated into (split-K atomics)
    ldq = dqkv.stride(0)
    st = _stream()
    # dV[j][dd] = sum_i P[i][j] dres[i][dd]
    w = WgradDesc()
    w.M, w.N, w.batch, w.bdiv = S, dk, n * heads, heads
    w.images, w.Hk, w.Wk = 1, S, 1
    plain_rows(w.g, P[0], rows=S)
    w.g.ld0 = Sp
    w.g_bs0, w.g_bs1 = heads * S * Sp, S * Sp
    plain_rows(w.x, dres, rows=S)
    w.x_bs0, w.x_bs1 = S * dres.stride(0), dk
    w.g_stride = w.x_stride = 1
    w.ntaps = 1
    w.dw, w.ldw = dqkv.data_ptr() + 8 * dk, ldq
    w.w_bs0, w.w_bs1 = S * ldq, 3 * dk
    w.alpha, w.accumulate = 1.0, 2
    lib.call("sdm_wgrad", C.byref(w), st)
    # dP[i][j] = sum_dd dres[i][dd] v[j][dd]
    dP = torch.empty_like(P)
    d = _base_desc(S, S, dk)
    d.batch, d.bdiv = n * heads, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, dres, rows=S)
    d.a_bs0, d.a_bs1 = S * dres.stride(0), dk
    d.b, d.ldb, d.b_kmajor = qkv.data_ptr() + 8 * dk, ld, 0
    d.b_bs0, d.b_bs1 = S * ld, 3 * dk
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(dP), Sp, heads * S * Sp, S * Sp
    _launch_igemm(d, dev)
    lib.call("sdm_col_softmax_bwd", _p(P), _p(dP), n * heads, S, Sp, float(dk) ** -0.5, st)
    # dQ[i][dd] = sum_j dS[i][j] k[j][dd]
    d = _base_desc(S, dk, S)
    d.batch, d.bdiv = n * heads, heads
    d.Ht, d.Wt = S, 1
    plain_rows(d.a, dP[0], rows=S)
    d.a.ld0 = Sp
    d.a_bs0, d.a_bs1 = heads * S * Sp, S * Sp
    d.b, d.ldb, d.b_kmajor = qkv.data_ptr() + 4 * dk, ld, 1
    d.b_bs0, d.b_bs1 = S * ld, 3 * dk
    d.ntaps[0] = 1
    d.Ho, d.Wo = S, 1
    d.out, d.ldo, d.o_bs0, d.o_bs1 = _p(dqkv), ldq, S * ldq, 3 * dk
    _launch_igemm(d, dev)
    # dK[j][dd] = sum_i dS[i][j] q[i][dd]
    w = WgradDesc()
    w.M, w.N, w.batch, w.bdiv = S, dk, n * heads, heads
    w.images, w.Hk, w.Wk = 1, S, 1
    plain_rows(w.g, dP[0], rows=S)
    w.g.ld0 = Sp
    w.g_bs0, w.g_bs1 = heads * S * Sp, S * Sp
    plain_rows(w.x, qkv, rows=S)
    w.x_bs0, w.x_bs1 = S * ld, 3 * dk
    w.g_stride = w.x_stride = 1
    w.ntaps = 1
    w.dw, w.ldw = dqkv.data_ptr() + 4 * dk, ldq
    w.w_bs0, w.w_bs1 = S * ldq, 3 * dk
    w.alpha, w.accumulate = 1.0, 2
    lib.call("sdm_wgrad", C.byref(w), st)
    return dqkv


# ---- pointwise / reductions -------------------------------------------------------------------
def stem_conv(x_nchw, wt, bias, act=ACT_SILU, save_z=False, out_dtype=torch.float32):
    n, cin, h, w = x_nchw.shape
    cout = wt.shape[1]
    if out_dtype == BF16:
        out = torch.empty((n * h * w, cout), dtype=BF16, device=wt.device)
        z = torch.empty_like(out) if save_z else None
        lib.call("sdm_stem_conv_bf16", _p(x_nchw), _p(wt), _p(bias), _p(out), _p(z), n, cin, h, w, cout, act, _stream())
        return Act(out, n, h, w, z=z)
    out = torch.empty((n * h * w, cout), dtype=torch.float32, device=wt.device)
    z = torch.empty_like(out) if save_z else None
    lib.call("sdm_stem_conv", _p(x_nchw), _p(wt), _p(bias), _p(out), _p(z), n, cin, h, w, cout, act, _stream())
    return Act(out, n, h, w, z=z)


HEAD_MFMA = [True]      # bf16 rows: head conv on the matrix cores (False: the one-wave-per-pixel kernel)
HEAD_STRIP = [os.environ.get("SDM_HEAD_STRIP", "1") != "0"]   # ... by the dedicated strip kernel (False: a tile of sdm_igemm_bf16)


def head_conv(x, wt, bias, act=ACT_NONE, wt16=None):
    """out_layers.1: rows [n*h*w, c] -> NCHW fp32 [n, c_out, h, w] (+Tanh).  bf16 rows with ``wt16`` (bf16 taps
    [9, c_out, c]): the contraction runs on sdm_igemm_bf16 with an fp32 result (c_out = 3 of a 64-wide tile: wasteful per
    FLOP, 3x faster than one wave per pixel at 128x128 / 256x256), then one pass lays the rows out as NCHW."""
    cout = wt.shape[1]
    out = torch.empty((x.n, cout, x.h, x.w), dtype=torch.float32, device=wt.device)
    if (x.t.dtype == BF16 and wt16 is not None and HEAD_MFMA[0] and HEAD_STRIP[0] and x.t2 is None and x.scale is None
            and x.c in (64, 128) and cout <= 16 and x.t.stride(0) % 8 == 0 and x.t.data_ptr() % 16 == 0):
        # the strip-walking matrix-core kernel (csrc/sdm_edge_bf16.hip): rows read once, NCHW written from the accumulators
        lib.call("sdm_head_conv_mfma_bf16", _p(x.t), x.t.stride(0), _p(wt16), _p(bias), _p(out), x.n, x.c, x.h, x.w, cout, act,
                 _stream())
        return out
    if x.t.dtype == BF16 and wt16 is not None and HEAD_MFMA[0] and x.c % 8 == 0:
        rows = torch.empty((x.rows, 4 * ((cout + 3) // 4)), dtype=torch.float32, device=wt.device)
        d = _base_desc(x.rows, cout, x.c)
        d.Ht, d.Wt = x.h, x.w
        _set_taps(d, 0, _C3_COLS)
        fill_rows(d.a, x)
        d.b, d.ldb, d.b_tap_stride, d.b_kmajor = _p(wt16), x.c, cout * x.c, 0
        d.Ho, d.Wo = x.h, x.w
        d.out, d.ldo = _p(rows), rows.stride(0)
        d.bias, d.act = _p(bias), act
        _launch_igemm(d, wt.device, bf16=True, out_f32=True)
        lib.call("sdm_rows_to_nchw", _p(rows), rows.stride(0), _p(out), x.n, cout, x.h, x.w, _stream())
        return out
    if x.t.dtype == BF16:
        lib.call("sdm_head_conv_bf16", _p(x.t), _p(wt), _p(bias), _p(out), x.n, x.c, x.h, x.w, cout, act, _stream())
        return out
    lib.call("sdm_head_conv", _p(x.t), _p(wt), _p(bias), _p(out), x.n, x.c, x.h, x.w, cout, act, _stream())
    return out


def gn_affine(y, gamma, beta, mod, groups=32, eps=1e-5, want_stats=True):
    """y: Act (plain).  Returns (scale, shift, mean, rstd); scale/shift are [n, C] fp32 (on bf16 rows mean / rstd are
    None unless ``want_stats``: only the training path's backward needs them)."""
    C_ = y.c
    dev = y.t.device
    scale = torch.empty((y.n, C_), dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    mns = 0 if (mod is None or mod.shape[0] == 1) else mod.stride(0)
    if y.t.dtype == BF16:
        mean = torch.empty((y.n, groups), dtype=torch.float32, device=dev) if want_stats else None
        rstd = torch.empty_like(mean) if want_stats else None
        if y.stats is not None:        # the producing conv's epilogue already summed the statistics: no pass over y
            lib.call("sdm_gn_finalize", _p(y.stats), y.n, y.h * y.w, C_, groups, eps, _p(gamma), _p(beta), _p(mod), mns,
                     _p(scale), _p(shift), _p(mean), _p(rstd), _stream())
            y.stats = None
            return scale, shift, mean, rstd
        lib.call("sdm_gn_affine_bf16", _p(y.t), y.t.stride(0), y.n, y.h * y.w, C_, groups, eps, _p(gamma), _p(beta), _p(mod),
                 mns, _p(scale), _p(shift), _p(mean), _p(rstd), _stream())
        return scale, shift, mean, rstd
    mean = torch.empty((y.n, groups), dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    mns = 0 if (mod is None or mod.shape[0] == 1) else mod.stride(0)
    lib.call("sdm_gn_affine", _p(y.t), y.t.stride(0), y.n, y.h * y.w, C_, groups, eps, _p(gamma), _p(beta), _p(mod), mns,
             _p(scale), _p(shift), _p(mean), _p(rstd), _stream())
    return scale, shift, mean, rstd


GN_COOP = [os.environ.get("SDM_GN_COOP", "1") != "0"]
# one-launch AdaGN backward (sdm_gn_bwd_coop): OFF by default.  Measured in the fp32 training step it is SLOWER than the three
# small kernels (7.2 -> 7.45 ms): its slabs must all be resident before any can finish, and the weight-gradient stream
# keeps the CUs full, so the big maps wait 118 us where the three kernels take 69; the small maps tie (17 vs 15 us).
GN_COOP_BWD = [os.environ.get("SDM_GN_COOP_BWD", "0") == "1"]
class _GnCoopState(_StreamKeyed):
    """Exchange state of sdm_gn_apply_coop / sdm_gn_bwd_coop (group sums + arrival counters; zero between launches): one per
    (device, launching stream), one per captured graph -- two concurrent users never share sums or counters."""

    def _new(self, device):
        lib.device_error_init()          # the bounded waits of those kernels report through the failure word
        return torch.zeros(lib.raw("sdm_gn_apply_coop_state_bytes")(), dtype=torch.uint8, device=device)

    def _prealloc(self, device_index):
        if not any(d == device_index for (d, _) in self.by_stream):
            return None
        return self._new(torch.device("cuda", device_index))

    def get(self, device):
        store, key = self._slot(device)
        st = store.get(key)
        if st is None:
            if torch.cuda.is_current_stream_capturing():
                raise lib.SdmError("sdm_gn_apply_coop state must be created by a warm-up call before graph capture")
            st = store[key] = self._new(device)
        keep(st)
        return st


_GN_COOP_STATE = _GnCoopState()


def _gn_coop_state(device):
    return _GN_COOP_STATE.get(device)


def gn_apply_fused(y, gamma, beta, mod, groups=32, eps=1e-5, residual=None, train=False):
    """out = s * GroupNorm(y) + s (+ cat(residual.t, residual.t2)) in ONE launch; y: a plain fp32 Act or a Deferred conv
    output, whose finishing pass (bias, activation) is folded in.  Returns a plain Act; with ``train`` (fp32 rows)
    (out, y, mean, rstd): the conv's own rows (activated, with ``z`` where the conv kept a pre-activation) and the GroupNorm
    statistics, i.e. what UNet_ConvBlock records for its backward."""
    C_ = y.c
    dev = gamma.device
    dt = y.dtype if isinstance(y, Deferred) else y.t.dtype
    out = torch.empty((y.n * y.h * y.w, C_), dtype=dt, device=dev)
    mns = 0 if (mod is None or mod.shape[0] == 1) else mod.stride(0)
    x0 = None if residual is None else residual.t
    x1 = None if residual is None else residual.t2
    if isinstance(y, Deferred):
        src, ld, raw, bias, act = y.ws, C_, 1, y.bias, y.act
    else:
        src, ld, raw, bias, act = y.t, y.t.stride(0), 0, None, ACT_NONE
    if train:
        if dt == BF16:
            raise lib.SdmError("gn_apply_fused(train=True): fp32 rows only")
        mean = torch.empty((y.n, groups), dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        if raw:
            yrows = y.out if y.out is not None else torch.empty((y.n * y.h * y.w, C_), dtype=dt, device=dev)
            ya = Act(yrows, y.n, y.h, y.w, z=y.z)
        else:
            yrows, ya = None, y
        lib.call("sdm_gn_apply_fused_train", _p(src), ld, raw, 1.0, _p(bias), act, y.n, y.h * y.w, C_, groups, eps, _p(gamma),
                 _p(beta), _p(mod), mns, _p(x0), x0.stride(0) if x0 is not None else 0, _p(x1),
                 x1.stride(0) if x1 is not None else 0, residual.c1 if x1 is not None else (1 << 30), _p(out), out.stride(0),
                 _p(ya.z) if raw else None, _p(yrows), yrows.stride(0) if raw else 0, _p(mean), _p(rstd), _stream())
        return Act(out, y.n, y.h, y.w), ya, mean, rstd
    if GN_COOP[0]:
        # slab-per-workgroup form with ONE exchange of the group sums (sdm_gn_coop.hip) where a group is a thin column
        # of many rows; it declines (NOT_APPLICABLE) the shapes the one-workgroup-per-group kernel below is right for
        st = _gn_coop_state(dev)
        rc = lib.call("sdm_gn_apply_coop", _p(src), ld, raw, 1.0, _p(bias), act, y.n, y.h * y.w, C_, groups, eps, _p(gamma),
                      _p(beta), _p(mod), mns, _p(x0), x0.stride(0) if x0 is not None else 0, _p(x1),
                      x1.stride(0) if x1 is not None else 0, residual.c1 if x1 is not None else (1 << 30), _p(out),
                      out.stride(0), int(dt == BF16), _p(st), st.numel(), _stream())
        if rc != lib.NOT_APPLICABLE:
            return Act(out, y.n, y.h, y.w)
    lib.call("sdm_gn_apply_fused_bf16" if dt == BF16 else "sdm_gn_apply_fused", _p(src), ld, raw, 1.0, _p(bias), act, y.n, y.h * y.w, C_, groups, eps, _p(gamma), _p(beta),
             _p(mod), mns, _p(x0), x0.stride(0) if x0 is not None else 0, _p(x1), x1.stride(0) if x1 is not None else 0,
             residual.c1 if x1 is not None else (1 << 30), _p(out), out.stride(0), _stream())
    return Act(out, y.n, y.h, y.w)


def affine_residual(y, scale, shift, x):
    """out = scale*y + shift + cat(x.t, x.t2)  (plain Act); x None = affine only."""
    ans = 0 if (scale is None or scale.shape[0] == 1) else scale.stride(0)
    if y.t.dtype == BF16:
        out = torch.empty((y.rows, y.c), dtype=BF16, device=y.t.device)
        x0 = None if x is None else x.t
        x1 = None if x is None else x.t2
        lib.call("sdm_affine_residual_bf16", _p(y.t), y.t.stride(0), _p(scale), _p(shift), ans, _p(x0),
                 x0.stride(0) if x0 is not None else 0, _p(x1), x1.stride(0) if x1 is not None else 0,
                 x.c1 if x1 is not None else (1 << 30), _p(out), out.stride(0), y.n, y.h * y.w, y.c, _stream())
        return Act(out, y.n, y.h, y.w)
    out = torch.empty((y.rows, y.c), dtype=torch.float32, device=y.t.device)
    x0 = None if x is None else x.t
    x1 = None if x is None else x.t2
    lib.call("sdm_affine_residual", _p(y.t), y.t.stride(0), _p(scale), _p(shift), ans, _p(x0),
             x0.stride(0) if x0 is not None else 0, _p(x1), x1.stride(0) if x1 is not None else 0,
             x.c1 if x1 is not None else (1 << 30), _p(out), out.stride(0), y.n, y.h * y.w, y.c, _stream())
    return Act(out, y.n, y.h, y.w)


def gn_adagn_bwd(dout, y, z, n, hw, gamma, beta, mod, mean, rstd, dgamma, dbeta, groups=32, dmod=None):
    """Returns (dz [rows, C], dmod [n, C] or None).  ``dmod`` may be a preallocated contiguous [n, C] view."""
    C_ = y.shape[1]
    dev = y.device
    dz = torch.empty((n * hw, C_), dtype=y.dtype, device=dev)
    if dmod is None and mod is not None:
        dmod = torch.empty((n, C_), dtype=torch.float32, device=dev)
    ws = GNWS.get(n * C_ * 2 * 4, dev)         # persistent, zero-filled; the kernels leave it zero
    mns = 0 if (mod is None or mod.shape[0] == 1) else mod.stride(0)
    if GN_COOP_BWD[0]:
        # one launch instead of three (sdm_gn_coop.hip); declines maps that do not fit its register-resident slabs
        st = _gn_coop_state(dev)
        rc = lib.call("sdm_gn_bwd_coop", _p(dout), dout.stride(0), _p(y), y.stride(0), _p(z), z.stride(0) if z is not None else 0,
                      n, hw, C_, groups, _p(gamma), _p(beta), _p(mod), mns, _p(mean), _p(rstd), _p(dz), dz.stride(0), _p(dmod),
                      _p(dgamma), _p(dbeta), _p(ws), int(y.dtype == BF16), _p(st), st.numel(), _stream())
        if rc != lib.NOT_APPLICABLE:
            return dz, dmod
    coef = torch.empty((n * 3 * C_,), dtype=torch.float32, device=dev)
    lib.call("sdm_gn_adagn_bwd_bf16" if y.dtype == BF16 else "sdm_gn_adagn_bwd", _p(dout), dout.stride(0), _p(y), y.stride(0), _p(z), z.stride(0) if z is not None else 0,
             n, hw, C_, groups, _p(gamma), _p(beta), _p(mod), mns, _p(mean), _p(rstd), _p(dz), dz.stride(0), _p(dmod),
             _p(dgamma), _p(dbeta), _p(ws), _p(coef), _stream())
    return dz, dmod


def sinusoid(t, freq, D):
    out = torch.empty((t.shape[0], D), dtype=torch.float32, device=freq.device)
    lib.call("sdm_sinusoid", _p(t), _p(freq), t.shape[0], D, _p(out), _stream())
    return out


def linear_small(x, W, b, act=ACT_NONE, add=None, save_z=False):
    """``add`` (optional) has M rows, or ONE row that is broadcast over the batch (time embedding of a scalar t added to
    per-image label embeddings, custom_layers.py:97-98)."""
    M, K = x.shape
    O = W.shape[0]
    ldadd = 0
    if add is not None:
        if add.shape[0] not in (1, M) or add.shape[1] != O:
            raise lib.SdmError(f"linear_small: addend of shape {tuple(add.shape)} does not broadcast to ({M}, {O})")
        ldadd = add.stride(0) if add.shape[0] == M else 0
    if M >= 32 and not save_z and K % 4 == 0 and x.stride(0) % 4 == 0:
        # one-wave-per-feature GEMV kernel is for the small batches of sampling / batch-4 training; from a few dozen
        # rows on (guided sampling at batch 64-128) the matrix-core path is faster (141 us -> ~10 us at M=128)
        return linear_rows(x, W, b, act, res=add, ldres=None if add is None else ldadd)
    out = torch.empty((M, O), dtype=torch.float32, device=x.device)
    z = torch.empty_like(out) if save_z else None
    lib.call("sdm_linear_small", _p(x), x.stride(0), _p(W), _p(b), _p(add), ldadd, 0 if add is None else add.shape[0], _p(out),
             out.stride(0), _p(z), M, K, O, act, _stream())
    return (out, z) if save_z else out


def linear_small_bwd(dy, x, W, dx=None, dx_accumulate=False, dW=None, db=None):
    M, O = dy.shape
    K = W.shape[1]
    lib.call("sdm_linear_small_bwd", _p(dy), dy.stride(0), _p(x), x.stride(0) if x is not None else 0, _p(W), _p(dx),
             dx.stride(0) if dx is not None else 0, int(dx_accumulate), _p(dW), _p(db), M, K, O, _stream())


def colsum(x, out, accumulate=True):
    if x.dtype == BF16:
        if not accumulate:
            raise lib.SdmError("colsum on bf16 rows accumulates (hand in a zero-filled buffer)")
        lib.call("sdm_colsum_bf16", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _stream())
        return
    lib.call("sdm_colsum", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), int(accumulate), _stream())


def silu(x):
    x = x.contiguous()
    y = torch.empty_like(x)
    lib.call("sdm_silu", _p(x), _p(y), x.numel(), _stream())
    return y


def silu_bwd(dy, z, dx=None):
    if dx is None:
        dx = torch.empty_like(dy)
    lib.call("sdm_silu_bwd_bf16" if dy.dtype == BF16 else "sdm_silu_bwd", _p(dy), _p(z), _p(dx), dy.numel(), _stream())
    return dx


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    lib.call("sdm_add", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def mse(a, b, want_grad=True):
    loss = torch.empty((1,), dtype=torch.float32, device=a.device)
    grad = torch.empty_like(a) if want_grad else None
    lib.call("sdm_mse", _p(a), _p(b), _p(loss), _p(grad), a.numel(), _stream())
    return loss, grad


def tanh_bwd(dy, y):
    dx = torch.empty_like(dy)
    lib.call("sdm_tanh_bwd", _p(dy), _p(y), _p(dx), dy.numel(), _stream())
    return dx


def q_sample(x0, eps, alpha_bar):
    xt = torch.empty_like(x0)
    lib.call("sdm_q_sample", _p(x0), _p(eps), _p(alpha_bar), _p(xt), x0.shape[0], x0[0].numel(), _stream())
    return xt


def ddpm_update(x, eps, z, coef, seed=0, rng_step=None, cursor=None, nsteps=0):
    """z: injected noise or None; rng_step (device int64 word, e.g. the step's t_cur): draw z in the kernel (Philox);
    cursor: the sampler's step cursor, advanced by this launch."""
    lib.call("sdm_ddpm_update", _p(x), _p(eps), _p(z), _p(coef), x.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(rng_step), _p(cursor),
             nsteps, _stream())


def ddim_update(x, eps, coef, x0_out=None, cursor=None, nsteps=0):
    lib.call("sdm_ddim_update", _p(x), _p(eps), _p(coef), _p(x0_out), x.numel(), _p(cursor), nsteps, _stream())


def cfg_combine(eps_2n, w):
    """eps_2n: [2N, ...] = [conditional | unconditional]; returns e_u + w (e_c - e_u) of shape [N, ...]."""
    n = eps_2n.shape[0] // 2
    out = torch.empty((n,) + tuple(eps_2n.shape[1:]), dtype=torch.float32, device=eps_2n.device)
    lib.call("sdm_cfg_combine", _p(eps_2n), _p(out), float(w), out.numel(), _stream())
    return out


def cold_update(x, x0_hat, noise, coef, x0_keep=None, cursor=None, nsteps=0):
    lib.call("sdm_cold_update", _p(x), _p(x0_hat), _p(noise), _p(coef), x.numel(), _p(x0_keep), _p(cursor), nsteps, _stream())


def area_resize(x, size):
    """F.interpolate(x, size=size, mode="area") for NCHW fp32 on the device."""
    n, c, h, w = x.shape
    ho, wo = size
    x = x.float().contiguous()
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    lib.call("sdm_area_resize", _p(x), _p(out), n * c, h, w, ho, wo, _stream())
    return out


BANK_MFMA = [os.environ.get("SDM_BANK_MFMA", "1") != "0"]


class LinearBank:
    """The y_scale projections of every AdaGN in the net, evaluated in ONE launch (sdm_linear_multi) and
    differentiated w.r.t. the shared embedding in ONE launch (sdm_linear_multi_bwd_dx).  Holds the device job
    tables; rebuilt when a parameter pointer or the embedding batch changes."""

    MAX_SETS = 8                   # table sets kept (LRU); a live graph pins the set it captured through capture_scope
    _all = None                    # weak set of the live banks (capture_scope sizes a set of its own for each)

    def __init__(self):
        import weakref
        if LinearBank._all is None:
            LinearBank._all = weakref.WeakSet()
        LinearBank._all.add(self)
        self.key = None
        self.built = {}            # key -> table set, in least-recently-used order

    _FIELDS = ("sizes", "out", "dout", "views", "dviews", "jobs", "bjobs", "chunk_job", "chunk_o0", "maxO")

    @staticmethod
    def _user():
        """Who the output buffers of a table set belong to: the launching stream, or the capture in progress -- like every
        other single-user buffer of the library (_StreamKeyed).  Two forwards of one net on two streams write their
        modulation rows into different buffers; a captured graph has a set of its own (made by capture_scope)."""
        scope = capture_scope.current()
        return ("capture", scope.serial) if scope is not None else ("stream", _stream())

    def _store(self, key, linears):
        self.built[key] = {k: getattr(self, k) for k in self._FIELDS}
        self.built[key]["linears"] = list(linears)
        while len(self.built) > self.MAX_SETS:            # evict the least recently used set; a graph that
            self.built.pop(next(iter(self.built)))        # captured it keeps it alive until the graph is destroyed

    def prealloc_for_capture(self, scope):
        """A table set for the capture that is about to start: same batch and projections as the most recent forward (the
        capture's warm-up run), buffers and job tables of its own."""
        if self.key is None or self.key not in self.built:
            return None
        cur = self.built[self.key]
        linears = cur["linears"]
        saved = {k: getattr(self, k) for k in self._FIELDS}
        self._build(linears, self.key[0], self.key[1])
        key = self.key[:4] + (("capture", scope.serial),)
        self._store(key, linears)
        for k, v in saved.items():
            setattr(self, k, v)
        return self.built[key]

    def _build(self, linears, M, dev):
        import ctypes
        self.sizes = [l.weight.shape[0] for l in linears]
        total = sum(self.sizes)
        self.out = torch.empty(M * total, dtype=torch.float32, device=dev)
        self.dout = torch.zeros(M * total, dtype=torch.float32, device=dev)
        self.views, self.dviews = [], []
        jobs = (lib.LinearJob * len(linears))()
        bjobs = (lib.LinearJob * len(linears))()
        cj, co, off = [], [], 0
        for j, l in enumerate(linears):
            O = self.sizes[j]
            v = self.out[off:off + M * O].view(M, O)
            dv = self.dout[off:off + M * O].view(M, O)
            self.views.append(v)
            self.dviews.append(dv)
            jobs[j].w, jobs[j].b, jobs[j].out, jobs[j].O = l.weight.data_ptr(), l.bias.data_ptr(), v.data_ptr(), O
            bjobs[j].w, bjobs[j].b, bjobs[j].out, bjobs[j].O = l.weight.data_ptr(), 0, dv.data_ptr(), O
            for o0 in range(0, O, 64):
                cj.append(j)
                co.append(o0)
            off += M * O
        self.jobs = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.bjobs = torch.frombuffer(bytearray(bytes(bjobs)), dtype=torch.uint8).to(dev)
        self.chunk_job = torch.tensor(cj, dtype=torch.int32, device=dev)
        self.chunk_o0 = torch.tensor(co, dtype=torch.int32, device=dev)
        self.maxO = max(self.sizes)

    def forward(self, linears, emb, private=False):
        M, K = emb.shape
        key = (M, emb.device, tuple(l.weight.data_ptr() for l in linears), tuple(l.bias.data_ptr() for l in linears),
               self._user())
        if key != self.key:
            if key not in self.built:
                if torch.cuda.is_current_stream_capturing():
                    raise lib.SdmError("y_scale job tables must be built by a warm-up call before graph capture")
                self._build(linears, M, emb.device)
                self._store(key, linears)
            else:
                self.built[key] = self.built.pop(key)             # most recently used last
                for k in self._FIELDS:
                    setattr(self, k, self.built[key][k])
            self.key = key
        keep(self.built[key])
        if M >= 64 and BANK_MFMA[0] and emb.is_contiguous():
            # large batches (C3: 128 U-Net images): one matrix-core GEMM per projection instead of the small-M VALU kernel
            # (0.55 ms of a 27.5 ms forward); at 32 rows (C4 training) the 20 extra launches cost more than they save
            for l, v in zip(linears, self.views):
                linear_rows(emb, l.weight, l.bias, out=v)
        else:
            lib.call("sdm_linear_multi", _p(emb), emb.stride(0), M, K, _p(self.jobs), len(linears), self.maxO, _stream())
        if not private:
            return self.views
        # training: the modulation rows are read again by backward, possibly after another forward has reused the
        # bank's output buffer -> hand out views of a private copy (one device-to-device copy)
        flat, views, off = self.out.clone(), [], 0
        for O in self.sizes:
            views.append(flat[off:off + M * O].view(M, O))
            off += M * O
        return views

    def tables(self):
        """The table set of the most recent forward (keep it with the forward's context: a later forward with another
        embedding batch switches the bank to a different set)."""
        return self.built[self.key]

    @staticmethod
    def backward_dx(tables, demb):
        """demb[m][k] += sum over all projections of dmod_j @ W_j (dmod_j were written into tables['dviews'])."""
        M, K = demb.shape
        lib.call("sdm_linear_multi_bwd_dx", _p(tables["bjobs"]), _p(tables["chunk_job"]), _p(tables["chunk_o0"]),
                 tables["chunk_job"].numel(), _p(demb), demb.stride(0), M, K, _stream())
