"""ctypes binding of libsdm_hip.so (C ABI declared in include/sdm_hip.h).

There is deliberately no fallback: if the library is missing the import fails, and every
launch wrapper raises on a non-zero return code.  Nothing here computes anything.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDM_HIP_LIB") or os.path.join(_HERE, "libsdm_hip.so")     # override: kernel experiments only

MAX_PHASE, MAX_TAPS = 4, 16
ACT_NONE, ACT_SILU, ACT_TANH, ACT_EXP = 0, 1, 2, 3

fp = C.c_void_p  # device pointers travel as integers


class Rows(C.Structure):
    _fields_ = [("p0", fp), ("p1", fp), ("csplit", C.c_int), ("ld0", C.c_int64), ("ld1", C.c_int64),
                ("scale", fp), ("shift", fp), ("aff_nstride", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class IgemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batch", C.c_int), ("bdiv", C.c_int),
        ("a", Rows), ("a_bs0", C.c_int64), ("a_bs1", C.c_int64),
        ("Ht", C.c_int), ("Wt", C.c_int), ("sstride", C.c_int), ("nphase", C.c_int),
        ("ntaps", C.c_int * MAX_PHASE),
        ("tap_dy", (C.c_int8 * MAX_TAPS) * MAX_PHASE),
        ("tap_dx", (C.c_int8 * MAX_TAPS) * MAX_PHASE),
        ("tap_w", (C.c_uint8 * MAX_TAPS) * MAX_PHASE),
        ("b", fp), ("ldb", C.c_int64), ("b_tap_stride", C.c_int64), ("b_kmajor", C.c_int),
        ("b_bs0", C.c_int64), ("b_bs1", C.c_int64),
        ("ostride", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("ph_oy", C.c_int * MAX_PHASE), ("ph_ox", C.c_int * MAX_PHASE),
        ("out", fp), ("ldo", C.c_int64), ("o_bs0", C.c_int64), ("o_bs1", C.c_int64),
        ("alpha", C.c_float), ("bias", fp), ("act", C.c_int), ("out_preact", fp),
        ("res", fp), ("ldres", C.c_int64), ("r_bs0", C.c_int64), ("r_bs1", C.c_int64),
        ("res2", fp), ("ldres2", C.c_int64),
        ("dact_z", fp), ("ldz", C.c_int64), ("dact_ncols", C.c_int),
        ("ws", fp), ("ws_bytes", C.c_size_t), ("gn_sums", fp),
        ("col_stats", fp), ("col_stats_bs", C.c_int64), ("col_shift", fp), ("col_mul", fp), ("colvec_bs", C.c_int64),
        ("defer_epilogue", C.c_int),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("images", C.c_int), ("Hk", C.c_int), ("Wk", C.c_int),
        ("batch", C.c_int), ("bdiv", C.c_int),
        ("g", Rows), ("g_stride", C.c_int), ("x", Rows), ("x_stride", C.c_int),
        ("g_bs0", C.c_int64), ("g_bs1", C.c_int64), ("x_bs0", C.c_int64), ("x_bs1", C.c_int64),
        ("ntaps", C.c_int),
        ("g_dy", C.c_int8 * MAX_TAPS), ("g_dx", C.c_int8 * MAX_TAPS),
        ("x_dy", C.c_int8 * MAX_TAPS), ("x_dx", C.c_int8 * MAX_TAPS),
        ("tap_w", C.c_uint8 * MAX_TAPS),
        ("dw", fp), ("ldw", C.c_int64), ("w_tap_stride", C.c_int64), ("w_bs0", C.c_int64), ("w_bs1", C.c_int64),
        ("dbias", fp), ("alpha", C.c_float), ("accumulate", C.c_int),
    ]


class LinearJob(C.Structure):
    _fields_ = [("w", fp), ("b", fp), ("out", fp), ("O", C.c_int), ("pad", C.c_int)]


class AdamJob(C.Structure):
    _fields_ = [("p", fp), ("g", fp), ("m", fp), ("v", fp), ("n", C.c_int64),
                ("w16", fp), ("w16t", fp), ("R", C.c_int), ("Cc", C.c_int), ("ldd", C.c_int), ("reserved", C.c_int)]


class SdmError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
        "There is no CPU or PyTorch fallback for the U-Net kernels.")

_lib = C.CDLL(LIB_PATH)

_i, _i64, _f, _sz = C.c_int, C.c_int64, C.c_float, C.c_size_t
_PROTOS = {
    "sdm_version": (C.c_int, []),
    "sdm_last_error": (C.c_char_p, []),
    "sdm_device_info": (C.c_int, [C.POINTER(C.c_int)]),
    "sdm_struct_size": (_sz, [_i]),
    "sdm_stream_order": (_i, [fp, fp]),
    "sdm_launch_count": (C.c_ulonglong, []),
    "sdm_plan_begin": (_i, []),
    "sdm_plan_end": (_i, []),
    "sdm_plan_launches": (_i, [_i]),
    "sdm_plan_run": (_i, [_i, _i, fp]),
    "sdm_plan_free": (_i, [_i]),
    "sdm_prof_begin": (_i, [_i]),
    "sdm_prof_bracket_ms": (C.c_double, [_i, fp]),
    "sdm_prof_end": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sdm_igemm": (_i, [C.POINTER(IgemmDesc), fp]),
    "sdm_igemm_workspace_bytes": (_sz, [C.POINTER(IgemmDesc)]),
    "sdm_wgrad": (_i, [C.POINTER(WgradDesc), fp]),
    "sdm_stem_conv": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv": (_i, [fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv_dgrad": (_i, [fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_edge_wgrad_workspace_bytes": (_sz, [_i, _i]),
    "sdm_stem_conv_wgrad": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv_wgrad": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_nchw_to_rows": (_i, [fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_gn_affine": (_i, [fp, _i64, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, fp, fp, fp, fp]),
    "sdm_gn_apply_fused": (_i, [fp, _i64, _i, _f, fp, _i, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, _i64, fp, _i64, _i, fp, _i64, fp]),
    "sdm_gn_apply_coop_state_bytes": (_sz, []),
    "sdm_device_error_init": (_i, []),
    "sdm_device_error_poll": (_i, []),
    "sdm_gn_apply_coop": (_i, [fp, _i64, _i, _f, fp, _i, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, _i64, fp, _i64, _i, fp, _i64, _i, fp, _sz,
                              fp]),
    "sdm_gn_bwd_coop": (_i, [fp, _i64, fp, _i64, fp, _i64, _i, _i, _i, _i, fp, fp, fp, _i, fp, fp, fp, _i64, fp, fp, fp, fp, _i, fp,
                            _sz, fp]),
    "sdm_gn_apply_fused_train": (_i, [fp, _i64, _i, _f, fp, _i, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, _i64, fp, _i64, _i, fp, _i64,
                                      fp, fp, _i64, fp, fp, fp]),
    "sdm_affine_residual": (_i, [fp, _i64, fp, fp, _i, fp, _i64, fp, _i64, _i, fp, _i64, _i, _i, _i, fp]),
    "sdm_gn_adagn_bwd": (_i, [fp, _i64, fp, _i64, fp, _i64, _i, _i, _i, _i, fp, fp, fp, _i, fp, fp, fp, _i64,
                              fp, fp, fp, fp, fp, fp]),
    "sdm_col_softmax": (_i, [fp, _i, _i, _i64, fp]),
    "sdm_col_softmax_bwd": (_i, [fp, fp, _i, _i, _i64, _f, fp]),
    "sdm_attn_scores_small": (_i, [fp, _i64, _i, _i, _i, _i, _f, fp, _i64, _i, fp]),
    "sdm_attn_small_nograd_workspace_bytes": (_sz, [_i, _i]),
    "sdm_attn_small_nograd": (_i, [fp, _i64, _i, _i, _i, _i, _f, fp, _i64, _i, fp, _sz, fp]),
    "sdm_attn_pv_small": (_i, [fp, _i64, fp, _i64, _i, _i, _i, _i, fp, _i64, _i, fp]),
    "sdm_sinusoid": (_i, [fp, fp, _i, _i, fp, fp]),
    "sdm_linear_small": (_i, [fp, _i64, fp, fp, fp, _i64, _i, fp, _i64, fp, _i, _i, _i, _i, fp]),
    "sdm_linear_multi": (_i, [fp, _i64, _i, _i, fp, _i, _i, fp]),
    "sdm_linear_multi_bwd_dx": (_i, [fp, fp, fp, _i, fp, _i64, _i, _i, fp]),
    "sdm_linear_small_bwd": (_i, [fp, _i64, fp, _i64, fp, fp, _i64, _i, fp, fp, _i, _i, _i, fp]),
    "sdm_colsum": (_i, [fp, _i64, _i, _i, fp, _i, fp]),
    "sdm_silu": (_i, [fp, fp, _sz, fp]),
    "sdm_silu_bwd": (_i, [fp, fp, fp, _sz, fp]),
    "sdm_add": (_i, [fp, fp, fp, _sz, fp]),
    "sdm_mse": (_i, [fp, fp, fp, fp, _sz, fp]),
    "sdm_tanh_bwd": (_i, [fp, fp, fp, _sz, fp]),
    "sdm_q_sample": (_i, [fp, fp, fp, fp, _i, _sz, fp]),
    "sdm_ddpm_update": (_i, [fp, fp, fp, fp, _sz, C.c_ulonglong, fp, fp, _i, fp]),
    "sdm_ddim_update": (_i, [fp, fp, fp, fp, _sz, fp, _i, fp]),
    "sdm_cfg_combine": (_i, [fp, fp, _f, _sz, fp]),
    "sdm_cold_update": (_i, [fp, fp, fp, fp, _sz, fp, fp, _i, fp]),
    "sdm_area_resize": (_i, [fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_step_begin": (_i, [fp, fp, fp, _i, _i, fp, fp, fp, fp, fp, _i, _i, fp]),
    "sdm_step_advance": (_i, [fp, _i, fp]),
    "sdm_u8_gather_nchw": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "sdm_adam_multi": (_i, [fp, _i, _i64, _f, _f, _f, _f, _f, _f, _f, fp, fp]),
    # bf16 storage path (forward / sampling)
    "sdm_igemm_bf16": (_i, [C.POINTER(IgemmDesc), _i, fp]),
    "sdm_igemm_bf16_workspace_bytes": (_sz, [C.POINTER(IgemmDesc)]),
    "sdm_igemm_bf16_tile_m": (_i, [C.POINTER(IgemmDesc)]),
    "sdm_attn_colstats_combine": (_i, [fp, _i, _i, _i, fp, fp, fp]),
    "sdm_cast_bf16": (_i, [fp, fp, _sz, fp]),
    "sdm_stem_conv_bf16": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv_bf16": (_i, [fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv_mfma_bf16": (_i, [fp, C.c_int64, fp, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "sdm_gn_affine_bf16": (_i, [fp, _i64, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, fp, fp, fp, fp]),
    "sdm_affine_residual_bf16": (_i, [fp, _i64, fp, fp, _i, fp, _i64, fp, _i64, _i, fp, _i64, _i, _i, _i, fp]),
    "sdm_col_softmax_bf16": (_i, [fp, fp, _i, _i, _i64, _i64, fp]),
    "sdm_gn_apply_fused_bf16": (_i, [fp, _i64, _i, _f, fp, _i, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, _i64, fp, _i64, _i, fp, _i64, fp]),
    "sdm_gn_finalize": (_i, [fp, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, fp, fp, fp, fp]),
    "sdm_rows_to_nchw": (_i, [fp, _i64, fp, _i, _i, _i, _i, fp]),
    "sdm_transpose_bf16": (_i, [fp, _i64, _i64, _i64, _i, fp, _i64, _i64, _i, _i, _i, fp]),
    # bf16 training path
    "sdm_wgrad_bf16": (_i, [C.POINTER(WgradDesc), fp]),
    "sdm_cast_bf16_t": (_i, [fp, fp, _i, _i, _i, _i, fp]),
    "sdm_gn_adagn_bwd_bf16": (_i, [fp, _i64, fp, _i64, fp, _i64, _i, _i, _i, _i, fp, fp, fp, _i, fp, fp, fp, _i64,
                                   fp, fp, fp, fp, fp, fp]),
    "sdm_col_softmax_bwd_bf16": (_i, [fp, _i64, fp, _i64, fp, _i64, _i, _i, _f, fp]),
    "sdm_silu_bwd_bf16": (_i, [fp, fp, fp, _sz, fp]),
    "sdm_colsum_bf16": (_i, [fp, _i64, _i, _i, fp, fp]),
    "sdm_stem_conv_wgrad_bf16": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv_dgrad_bf16": (_i, [fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_head_conv_wgrad_bf16": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "sdm_nchw_to_rows_bf16": (_i, [fp, fp, _i, _i, _i, _i, _i, fp]),
    # data-parallel exchange (communicator owned by the library)
    "sdm_dp_available": (_i, []),
    "sdm_dp_unique_id": (_i, [C.c_char_p, _sz]),
    "sdm_dp_init": (_i, [C.c_char_p, _sz, _i, _i, C.POINTER(fp)]),
    "sdm_dp_rank": (_i, [fp]),
    "sdm_dp_world": (_i, [fp]),
    "sdm_dp_allreduce_bucket": (_i, [fp, fp, _i64, _i, _i, fp]),
    "sdm_dp_reserve": (_i, [fp, _i64]),
    "sdm_dp_async_error": (_i, [fp]),
    "sdm_dp_finalize": (_i, [fp]),
}
EXPORTS = tuple(_PROTOS)

for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(_lib, _name)       # AttributeError here == symbol missing from the .so
    _fn.restype, _fn.argtypes = _res, _args

for _which, _cls in enumerate((IgemmDesc, WgradDesc, Rows, LinearJob, AdamJob)):
    if _lib.sdm_struct_size(_which) != C.sizeof(_cls):
        raise ImportError(f"ABI mismatch for {_cls.__name__}: C {_lib.sdm_struct_size(_which)} vs ctypes {C.sizeof(_cls)}")


def last_error():
    return _lib.sdm_last_error().decode()


STATS_DONE = 2    # sdm_igemm_bf16(gn_sums=...) accumulated the GroupNorm statistics in its epilogue
NOT_APPLICABLE = 3   # sdm_gn_apply_coop: shape outside its envelope, nothing launched
DEFERRED = 1      # sdm_igemm(defer_epilogue=1) left raw split-K sums for the caller's next kernel to finish


def call(name, *args):
    rc = getattr(_lib, name)(*args)
    if rc < 0:
        raise SdmError(f"{name} failed (rc={rc}): {last_error()}")
    return rc


def raw(name):
    return getattr(_lib, name)


_DEVERR_READY = [False]


def device_error_init():
    """Allocate the host-mapped failure word the bounded device-side waits report through (once per process; must happen
    outside stream capture, so ops calls it before the first launch of a process)."""
    if not _DEVERR_READY[0]:
        call("sdm_device_error_init")
        _DEVERR_READY[0] = True


def check_device_error():
    """Raise SdmError if a kernel has reported a failure (exact after a stream synchronisation)."""
    call("sdm_device_error_poll")


def version():
    return _lib.sdm_version()


def launch_count():
    return int(_lib.sdm_launch_count())


PROF_CLASSES = ("igemm_128x128", "igemm_64x64", "igemm_32x128", "wgrad_128x128", "wgrad_64x64", "igemm_128x64", "igemm_skinny")


def prof_begin(capacity=100000):
    call("sdm_prof_begin", capacity)


def prof_bracket_ms(stream, reps=64):
    """Median time between the two events around an EMPTY kernel on ``stream`` (raw stream handle): the per-launch timing's
    own overhead plus the empty kernel's ~1-2 us."""
    return float(_lib.sdm_prof_bracket_ms(int(reps), stream))


def prof_end():
    k = len(PROF_CLASSES)
    ms = (C.c_double * k)()
    n = (C.c_longlong * k)()
    fl = (C.c_double * k)()
    by = (C.c_double * k)()
    call("sdm_prof_end", ms, n, fl, by)
    return {PROF_CLASSES[i]: dict(ms=ms[i], launches=n[i], flops=fl[i], bytes=by[i]) for i in range(k)}
