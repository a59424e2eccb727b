"""Route dorpatch_amd.ops through the host emulation library for the duration of a test.

TEST INFRASTRUCTURE ONLY.  The product path (dorpatch_amd.ops) refuses CPU tensors and only ever
loads libdorpatch_hip.so; this context manager swaps in tests/hipemu/libdorpatch_emu.so — the same
translation unit compiled as host C++ — and relaxes the "must be a GPU tensor" guards, so that the
Python host wrappers + the kernels' logic can be exercised on CPU tensors in the GPU-less container.
"""
import contextlib
import ctypes

import numpy as np
import torch

from dorpatch_amd import _lib, ops
from . import build_emu

_handle = None


def emu_lib():
    """ctypes handle of the emulation library with the product's prototypes, or None (no host clang++)."""
    global _handle
    if _handle is None:
        path = build_emu.build()
        if path is None:
            return None
        lib = ctypes.CDLL(path)
        for name, (restype, argtypes) in _lib.PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        assert lib.dp_abi_version() == _lib.DP_ABI_VERSION
        _handle = lib
    return _handle


def _chk_cpu(t, dtype, name):
    if not isinstance(t, torch.Tensor) or t.is_cuda:
        raise RuntimeError("hipemu: `%s` must be a CPU tensor" % name)
    if t.dtype != dtype:
        raise TypeError(f"`{name}` must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"`{name}` must be contiguous")
    return t


def _gn_supported(x, groups):
    if not (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() >= 2) or x.shape[1] % groups:
        return False
    L = (x.shape[1] // groups) * int(np.prod(x.shape[2:]))
    return L % 4 == 0 and L < (1 << 20)


def _pool_supported(x):
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0)


def _sub_supported(x):
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.is_contiguous())


def _stem_supported(x, weight, stride, padding):
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4
            and tuple(weight.shape[1:]) == (3, 7, 7) and tuple(stride) == (2, 2) and tuple(padding) == (3, 3)
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def _c1_supported(x, weight):
    if not (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    HW = x.shape[2] * x.shape[3]
    return (weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1) and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 16 == 0 and weight.shape[0] % 64 == 0 and (HW % 4 == 0 or HW == 49))


def _c3_supported(x, weight, stride=(1, 1), padding=(1, 1)):
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and x.shape[2] == x.shape[3] and x.shape[2] in ops.CONV3X3_SIDES and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 8 == 0 and weight.shape[0] % 64 == 0)


def _c3s2_supported(x, weight, stride=(2, 2), padding=(1, 1)):
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and x.shape[2] == x.shape[3] and x.shape[2] in ops.CONV3X3S2_SIDES and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 8 == 0 and weight.shape[0] % 64 == 0)


def _stemc_supported(x, weight, stride=(2, 2), padding=(3, 3)):
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape) == (64, 3, 7, 7) and tuple(stride) == (2, 2) and tuple(padding) == (3, 3)
            and x.shape[1] == 3 and x.shape[3] == 224 and x.shape[2] % 2 == 0)


def _c3s2b_supported(dy, weight, stride=(2, 2), padding=(1, 1)):
    return (isinstance(dy, torch.Tensor) and dy.dtype == torch.float32 and dy.dim() == 4 and dy.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and dy.shape[2] == dy.shape[3] and dy.shape[2] in ops.CONV3X3S2_BWD_SIDES and weight.shape[0] == dy.shape[1]
            and weight.shape[0] % 16 == 0 and weight.shape[1] % 64 == 0)


def _poisoned(fn):
    """torch.empty / torch.empty_like that hand out NaN (float) or a sentinel (integer) instead of whatever the
    allocator had: an output element a kernel forgets to write then reaches the comparison as NaN / garbage."""
    def make(*a, **k):
        t = fn(*a, **k)
        if t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype == torch.bool:
                t.fill_(True)
            elif t.dtype in (torch.uint8, torch.int8):
                t.fill_(0x5B)
            else:
                t.fill_(-123456789 if t.dtype != torch.int16 else -12345)
        return t
    return make


@contextlib.contextmanager
def emulated_ops():
    import os
    poison = os.environ.get("HIPEMU_POISON", "0") == "1"     # tests/test_kernels_asan.py sets it for its child run
    saved_empty = (torch.empty, torch.empty_like)
    if poison:
        torch.empty, torch.empty_like = _poisoned(torch.empty), _poisoned(torch.empty_like)
    try:
        with _emulated_ops() as lib:
            yield lib
    finally:
        torch.empty, torch.empty_like = saved_empty


@contextlib.contextmanager
def _emulated_ops():
    import os
    lib = emu_lib()
    assert lib is not None, "no host clang++: cannot build the emulation library"
    saved = dict(lib=_lib._lib, req=ops.require_gpu, chk=ops._chk, stream=ops._stream, gn=ops.gn_relu_supported,
                 pool=ops.pad_maxpool_supported, stem=ops.stem_dgrad_supported, sub=ops.subsample2_supported,
                 c1=ops.conv1x1_supported, c3=ops.conv3x3_supported, c3s2=ops.conv3x3s2_supported,
                 c3s2b=ops.conv3x3s2_bwd_supported, stemc=ops.stem_conv_supported)
    if os.environ.get("HIPEMU_MFMA_CONVS", "0") == "1":     # the matrix-core convolutions through the emulation too (slow)
        ops.conv1x1_supported, ops.conv3x3_supported = _c1_supported, _c3_supported
        ops.conv3x3s2_supported = _c3s2_supported
        ops.conv3x3s2_bwd_supported = _c3s2b_supported
        ops.stem_conv_supported = _stemc_supported
    _lib._lib = lib
    ops._chk = _chk_cpu
    ops.require_gpu = lambda t, what: t
    ops._stream = lambda: None
    ops.gn_relu_supported, ops.pad_maxpool_supported, ops.stem_dgrad_supported = _gn_supported, _pool_supported, _stem_supported
    ops.subsample2_supported = _sub_supported
    try:
        yield lib
    finally:
        _lib._lib = saved["lib"]
        ops._chk, ops._stream, ops.require_gpu = saved["chk"], saved["stream"], saved["req"]
        ops.gn_relu_supported, ops.pad_maxpool_supported, ops.stem_dgrad_supported = saved["gn"], saved["pool"], saved["stem"]
        ops.subsample2_supported = saved["sub"]
        ops.conv1x1_supported, ops.conv3x3_supported = saved["c1"], saved["c3"]
        ops.conv3x3s2_supported = saved["c3s2"]
        ops.conv3x3s2_bwd_supported = saved["c3s2b"]
        ops.stem_conv_supported = saved["stemc"]
