"""ctypes binding of libdorpatch_hip.so (the C ABI in include/dorpatch_hip.h).

There is deliberately NO fallback: if the shared library is missing this module
raises, and every op in ``dorpatch_amd.ops`` refuses non-GPU tensors.  The CPU
oracle under ``oracle/`` is test infrastructure and is never imported here.
"""
import ctypes
import os

# torch must be imported first: it maps its bundled libamdhip64 (SONAME
# libamdhip64.so.7); our library's NEEDED entry then resolves to that same
# runtime instance, so device pointers and streams are shared with PyTorch.
import torch  # noqa: F401

from .build import LIB_PATH

DP_ABI_VERSION = 12
DP_MAX_RECTS = 4
DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK, DP_DEBUG_UPDATE_VARIANT, DP_DEBUG_APPLY_ORDER, DP_DEBUG_AFFINE_GATHER = 1, 2, 3, 4   # dp_debug_set knobs
DP_DEBUG_CONV1X1_VARIANT = 5
DP_DEBUG_CONV3X3_VARIANT = 6
DP_S2BWD_PAIRS, DP_S2BWD_CLASSES = 0, 1      # dp_conv3x3s2_bwd forms

c_float_p = ctypes.c_void_p  # device pointers travel as integers
c_int_p = ctypes.c_void_p
c_stream = ctypes.c_void_p


class DpNorm(ctypes.Structure):
    """dp_norm_t"""
    _fields_ = [("enable", ctypes.c_int),
                ("mean", ctypes.c_float * 3),
                ("std", ctypes.c_float * 3),
                ("fill", ctypes.c_float)]


class DpUpdateCfg(ctypes.Structure):
    """dp_update_cfg_t"""
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("stage", ctypes.c_int), ("unit", ctypes.c_int), ("win", ctypes.c_int),
                ("do_update", ctypes.c_int), ("density", ctypes.c_float),
                ("clip_min", ctypes.c_float), ("clip_max", ctypes.c_float)]


# name -> (restype, argtypes); mirrors include/dorpatch_hip.h one to one
_I, _F, _P, _L = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64
PROTOTYPES = {
    "dp_abi_version": (_I, []),
    "dp_error_string": (ctypes.c_char_p, [_I]),
    "dp_debug_set": (_I, [_I, _I]),
    "dp_sumsq_nchunk": (_I, [_I]),
    "dp_sumsq_partials": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "dp_blend": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P]),
    "dp_apply_fwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P]),
    "dp_apply_bwd_nslab": (_I, [_I, _I, _I]),
    "dp_apply_bwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P]),
    "dp_sum_slabs": (_I, [_P, _I, _L, _P, _I, _P]),
    "dp_apply_affine_fwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P]),
    "dp_apply_affine_bwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P]),
    "dp_cw_loss": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "dp_local_variance": (_I, [_P, _I, _I, _I, _P, _P]),
    "dp_struct_ntile": (_I, [_I, _I]),
    "dp_struct_loss": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "dp_reduce_rows": (_I, [_P, _I, _I, _F, _P, _P]),
    "dp_mask_stats": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "dp_project_update": (_I, [ctypes.POINTER(DpUpdateCfg)] + [_P] * 18),
    "dp_argmax": (_I, [_P, _I, _I, _P, _P]),
    "dp_conv3x3_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dp_conv3x3_gn_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dp_conv3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dp_conv3x3_wino_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dp_stem_conv_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "dp_conv3x3s2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "dp_conv1x1_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dp_gn_stats": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "dp_gn_relu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "dp_gn_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dp_gn_relu_bwd_gather": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dp_pad_maxpool_fwd": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "dp_pad_maxpool_bwd": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "dp_stem_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "dp_stem_dgrad_reduce": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P]),
    "dp_subsample2": (_I, [_P, _L, _I, _I, _P, _P]),
    "dp_subsample2_add": (_I, [_P, _L, _I, _I, _P, _P]),
    "dp_event_create": (_I, [ctypes.POINTER(ctypes.c_void_p)]),
    "dp_event_destroy": (_I, [_P]),
    "dp_event_elapsed_ms": (_I, [_P, _P, ctypes.POINTER(ctypes.c_float)]),
    "dp_apply_fwd_timed": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P, _P, _P]),
    "dp_apply_affine_fwd_timed": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(DpNorm), _P, _P, _P,
                                       _P]),
}

_lib = None


def _count_hip_runtimes():
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
        return len(paths), sorted(paths)
    except OSError:
        return 1, []


def load():
    """Load (once) and return the ctypes handle. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "dorpatch_amd has no CPU/PyTorch fallback by design.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.dp_abi_version() != DP_ABI_VERSION:
        raise ImportError("libdorpatch_hip.so ABI version mismatch: rebuild the extension")
    n, paths = _count_hip_runtimes()
    if n > 1:
        raise ImportError("two HIP runtimes are mapped into this process (%s): device pointers "
                          "would not be shared with PyTorch" % ", ".join(paths))
    _lib = lib
    return lib


def check(err, what):
    if err != 0:
        msg = load().dp_error_string(err)
        raise RuntimeError("%s failed: hipError %d (%s)" % (what, err, msg.decode() if msg else "?"))
