"""ctypes binding of libr2hip.so (the C ABI declared in include/r2hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the ops
raises.  Build it with ``python -m r2_gaussian_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# R2HIP_LIB selects an experiment build (profiling ablations); the default is the product library
LIB_PATH = os.environ.get("R2HIP_LIB") or os.path.join(_HERE, "libr2hip.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)

R2_ABI_VERSION = 3
R2_ERR_INVALID = -10001
R2_ERR_ALLOC = -10002

_f, _i, _p, _fp = C.c_float, C.c_int, C.c_void_p, C.c_void_p

_SIGNATURES = {
    "r2_abi_version": (C.c_int, []),
    "r2_last_error": (C.c_char_p, []),
    "r2_raster_forward": (C.c_int, [ALLOC_FN, _p, ALLOC_FN, _p, ALLOC_FN, _p, _i, _i, _i, _fp, _fp, _fp, _f, _fp, _fp,
                                    _fp, _fp, _fp, _f, _f, _i, _i, _fp, _p, _i, _p]),
    "r2_raster_backward": (C.c_int, [_i, _i, _i, _i, _fp, _fp, _f, _fp, _fp, _fp, _fp, _fp, _f, _f, _p, _p, _p, _p,
                                     _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _p]),
    "r2_mark_visible": (C.c_int, [_i, _fp, _fp, _fp, _p, _p]),
    "r2_raster_forward_batch": (C.c_int, [ALLOC_FN, _p, ALLOC_FN, _p, ALLOC_FN, _p, _i, _i, _i, _i, _fp, _fp, _fp, _f, _fp,
                                          _fp, _fp, _fp, _f, _f, _i, _fp, _p, _i, _p]),
    "r2_raster_backward_batch": (C.c_int, [_i, _i, _i, _i, _i, _fp, _fp, _f, _fp, _fp, _fp, _fp, _f, _f, _p, _p, _p, _p,
                                           _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _p]),
    "r2_voxel_forward": (C.c_int, [ALLOC_FN, _p, ALLOC_FN, _p, ALLOC_FN, _p, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f,
                                   _fp, _fp, _fp, _f, _fp, _fp, _i, _fp, _p, _p, _p, _i, _p]),
    "r2_voxel_backward": (C.c_int, [_i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _fp, _fp, _f, _fp, _fp, _p, _p, _p,
                                    _p, _p, _p, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _p]),
    "r2_voxel_forward_slab": (C.c_int, [ALLOC_FN, _p, ALLOC_FN, _p, ALLOC_FN, _p, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _i,
                                        _fp, _fp, _fp, _f, _fp, _fp, _i, _fp, _p, _p, _p, _i, _p]),
    "r2_voxel_backward_slab": (C.c_int, [_i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _i, _fp, _fp, _f, _fp, _fp, _p, _p, _p,
                                         _p, _p, _p, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _p]),
    "r2_knn_dist2": (C.c_int, [_i, _fp, _fp, _p]),
    "r2_knn_workspace_bytes": (C.c_size_t, [_i]),
    "r2_knn_dist2_ws": (C.c_int, [_i, _fp, _fp, _p, C.c_size_t, _p]),
    "r2_tile_first_stats": (None, [C.POINTER(C.c_longlong), _i]),
    "r2_defer_count_control": (None, [_i]),
    "r2_defer_count_stats": (None, [C.POINTER(C.c_longlong), _i]),
    "r2_voxel_sticks_control": (None, [_i]),
    "r2_voxel_sticks_limits": (None, [C.c_longlong, C.c_longlong]),
    "r2_voxel_sticks_stats": (None, [C.POINTER(C.c_longlong), _i]),
    "r2_thread_release": (None, []),
    "r2_path_stat_count": (C.c_int, []),
    "r2_path_stat_name": (C.c_char_p, [_i]),
    "r2_path_stats": (C.c_int, [C.POINTER(C.c_longlong), _i, _i]),
    "r2_densify_stats": (C.c_int, [_i, _p, _fp, _fp, _fp, _fp, _p]),
    "r2_densify_scratch_bytes": (C.c_size_t, [_i]),
    "r2_densify_classify": (C.c_int, [_i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _f, _f, _p, _f, _f, _i, _f, _f, _p, _p, _p]),
    "r2_densify_emit": (C.c_int, [_i, _p, _p, _p, _fp, _fp, _fp, _fp, _f, _f, _f, _p, _f, _f, _i, _f, _f, _p, _p, _p, _p, _fp,
                                  _fp, _fp, _p]),
    "r2_loss_l1_ssim_scratch_floats": (C.c_size_t, [_i, _i]),
    "r2_loss_l1_ssim": (C.c_int, [_i, _i, _fp, _fp, _f, _f, _fp, _fp, _fp, _p]),
    "r2_loss_tv3d_scratch_floats": (C.c_size_t, [_i, _i, _i]),
    "r2_loss_tv3d": (C.c_int, [_i, _i, _i, _fp, _f, _fp, _fp, _fp, _p]),
    "r2_fdk_filter": (C.c_int, [_i, _i, _i, _fp, _fp, _f, _i, _f, _f, _f, _fp, _p]),
    "r2_fdk_backproject": (C.c_int, [_i, _i, _i, _fp, _fp, _i, _f, _i, _i, _i, _f, _f, _f, _f, _f, _f, _fp, _p]),
    "r2_profile_enable": (None, [C.c_ulonglong]),
    "r2_profile_stage_count": (C.c_int, []),
    "r2_profile_stage_name": (C.c_char_p, [_i]),
    "r2_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong), _i]),
    "r2_sync_wait_stats": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong), _i]),
    "r2_depth_hint_control": (None, [_i]),
    "r2_tile_first_control": (None, [_i]),
    "r2_profile_host": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), _i]),
    "r2_raster_state_offset": (C.c_longlong, [_i, _i, C.c_longlong, _i, _i, C.POINTER(C.c_int)]),
    "r2_voxel_state_offset": (C.c_longlong, [_i, _i, C.c_longlong, _i, _i, _i, C.POINTER(C.c_int)]),
}

_lib = None


class R2HipError(RuntimeError):
    pass


def lib():
    """Load libr2hip.so once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise R2HipError(
                "libr2hip.so not found at %s -- the HIP extension is required (no fallback). "
                "Run `python -m r2_gaussian_amd.build`." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.r2_abi_version() != R2_ABI_VERSION:
            raise R2HipError("libr2hip.so ABI %d != expected %d" % (L.r2_abi_version(), R2_ABI_VERSION))
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what):
    if rc < 0:
        msg = lib().r2_last_error()
        raise R2HipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))
    return rc


def stage_names():
    L = lib()
    return [L.r2_profile_stage_name(i).decode() for i in range(L.r2_profile_stage_count())]


def profile_enable(stages=None):
    """Enable HIP-event timing for the named stages (None = all, [] = off)."""
    names = stage_names()
    if stages is None:
        mask = (1 << len(names)) - 1
    else:
        mask = 0
        for s in stages:
            mask |= 1 << names.index(s)
    lib().r2_profile_enable(mask)


def path_stats(reset=False):
    """-> {name: count} of the chain counters (csrc/dispatch.hpp): which chain every forward took and why."""
    L = lib()
    n = L.r2_path_stat_count()
    buf = (C.c_longlong * n)()
    L.r2_path_stats(buf, n, int(reset))
    return {L.r2_path_stat_name(i).decode(): int(buf[i]) for i in range(n)}


def sync_wait_stats(reset=True):
    """-> (total microseconds the host busy-waited for num_rendered, number of waits)."""
    us, n = C.c_double(0.0), C.c_longlong(0)
    lib().r2_sync_wait_stats(C.byref(us), C.byref(n), int(reset))
    return us.value, n.value


def profile_host(reset=True):
    """-> (host microseconds per forward before its synchronisation wait, after it, number of forwards)."""
    a, b, n = C.c_double(0.0), C.c_double(0.0), C.c_longlong(0)
    lib().r2_profile_host(C.byref(a), C.byref(b), C.byref(n), int(reset))
    return a.value / max(n.value, 1), b.value / max(n.value, 1), n.value


def profile_read(reset=True):
    """-> {stage: (total_ms, launches)} for stages that ran."""
    L = lib()
    n = L.r2_profile_stage_count()
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    L.r2_profile_read(ms, cnt, int(reset))
    names = stage_names()
    return {names[i]: (ms[i], cnt[i]) for i in range(n) if cnt[i] > 0}
