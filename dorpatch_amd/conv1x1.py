"""Frozen 1x1 convolutions of the backbone: which LIBRARY kernel runs them, decided by measurement.

``north_star`` leaves the backbone's MFMA work to rocBLAS / MIOpen; this module only chooses between
the two for the 33 stride-1 1x1 convolutions of ResNetV2-50 (reference call sites ``attack.py:222,
247`` through timm).  Why it matters (``profiles/r01c_bench_cfg2_kernel_stats_timed.txt``): MIOpen's
immediate mode sends ~30 of those convolutions (mostly their backward-data) through NHWC
implicit-GEMM kernels wrapped in NCHW<->NHWC ``batched_transpose_*`` kernels and a zero-fill
(``SubTensorOpWithScalar1d``) — 9 % + 1.2 % of a step spent re-laying-out activations.  In NCHW a
stride-1 1x1 convolution IS a strided-batched GEMM on the tensors as they lie in HBM:

    forward        out[n] (O x HW) = W   (O x C) @ x[n]  (C x HW)
    input gradient dx[n]  (C x HW) = W^T (C x O) @ dy[n] (O x HW)

(``torch.bmm`` with a stride-0 batch of the weight -> rocBLAS/hipBLASLt, fp32 MFMA, no copies).
Neither is always faster, so each (direction, N, C, O, HW) is timed once with both at first use
(1 warm-up + ``CAL_ITERS`` launches each, HIP events) and the winner is cached for the process:
MIOpen's own find mode, extended by one candidate it does not have.  ``MODE``: ``"auto"`` (default),
``"gemm"``, ``"miopen"`` (the plain ``F.conv2d`` path, bypassing this module); the environment variable
``DORPATCH_CONV1X1`` sets it at import.  A measured choice can differ between runs when two routes tie,
which changes fp32 summation order only (like the reference's ``cudnn.benchmark = True``,
``utils.py:17``); pin a mode for run-to-run bit reproducibility.

Weights are frozen on this path (``DorPatch.generate`` freezes the backbone): there is no weight
gradient.
"""
import time

import torch
import torch.nn.functional as F

import os

MODE = os.environ.get("DORPATCH_CONV1X1", "auto")     # "auto" | "gemm" | "miopen"
if MODE not in ("auto", "gemm", "miopen"):
    raise ValueError("DORPATCH_CONV1X1 must be auto, gemm or miopen, got %r" % MODE)
CAL_ITERS = 3
_choice = {}          # (direction, N, C, O, HW, device) -> "gemm" | "miopen"
_timings = {}         # same key -> (gemm_ms, miopen_ms)


def _fwd_gemm(x, w4d, _=None):
    N, C, H, W = x.shape
    O = w4d.shape[0]
    return torch.bmm(w4d.view(1, O, C).expand(N, O, C), x.view(N, C, H * W)).view(N, O, H, W)


def _fwd_miopen(x, w4d, _=None):
    return F.conv2d(x, w4d)


def _bwd_gemm(dy, w4d, x=None):
    N, O, H, W = dy.shape
    C = w4d.shape[1]
    return torch.bmm(w4d.view(O, C).t().unsqueeze(0).expand(N, C, O), dy.view(N, O, H * W)).view(N, C, H, W)


def _bwd_miopen(dy, w4d, x):
    # exactly the call autograd makes for F.conv2d (input gradient only); `x` is passed for its shape —
    # MIOpen's backward-data never reads it, but ATen wants a dense tensor there
    return torch.ops.aten.convolution_backward(dy, x, w4d, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                               (True, False, False))[0]


_IMPL = {("fwd", "gemm"): _fwd_gemm, ("fwd", "miopen"): _fwd_miopen,
         ("bwd", "gemm"): _bwd_gemm, ("bwd", "miopen"): _bwd_miopen}


def _time_ms(fn, t, w4d, x):
    fn(t, w4d, x)                                # warm-up: kernel selection / code-object load
    if t.is_cuda:
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(CAL_ITERS):
            fn(t, w4d, x)
        stop.record()
        stop.synchronize()
        return start.elapsed_time(stop) / CAL_ITERS
    t0 = time.perf_counter()
    for _ in range(CAL_ITERS):
        fn(t, w4d, x)
    return (time.perf_counter() - t0) * 1e3 / CAL_ITERS


def _pick(direction, t, w4d, x):
    if MODE in ("gemm", "miopen"):
        return MODE
    O, C = w4d.shape[0], w4d.shape[1]
    key = (direction, t.shape[0], C, O, t.shape[2] * t.shape[3], str(t.device))
    algo = _choice.get(key)
    if algo is None:
        with torch.no_grad():
            ms_lib = _time_ms(_IMPL[(direction, "miopen")], t, w4d, x)
            try:        # the GEMM route is the optional candidate: any refusal or disagreement keeps MIOpen
                ms_gemm = _time_ms(_IMPL[(direction, "gemm")], t, w4d, x)
                a, b = _IMPL[(direction, "gemm")](t, w4d, x), _IMPL[(direction, "miopen")](t, w4d, x)
                tol = 1e-3 * float(b.abs().max()) + 1e-30
                if not bool(((a - b).abs() <= tol).all()):
                    ms_gemm = float("inf")
            except RuntimeError:
                ms_gemm = float("inf")
        algo = "gemm" if ms_gemm < ms_lib else "miopen"
        _choice[key], _timings[key] = algo, (ms_gemm, ms_lib)
    return algo


def _run(direction, t, w4d, x=None):
    return _IMPL[(direction, _pick(direction, t, w4d, x))](t, w4d, x)


class Conv1x1Function(torch.autograd.Function):
    """``conv2d(x, w)`` for a frozen (O, C, 1, 1) filter, stride 1, no padding, NCHW fp32."""

    @staticmethod
    def forward(ctx, x, w4d):
        x = x.contiguous()
        ctx.save_for_backward(w4d, x)       # x: a reference only (shape for MIOpen's backward-data), never re-read
        return _run("fwd", x, w4d)

    @staticmethod
    def backward(ctx, dy):
        w4d, x = ctx.saved_tensors
        return _run("bwd", dy.contiguous(), w4d, x), None


def applicable(conv, x):
    """A frozen, folded, bias-free 1x1/1 convolution on an fp32 NCHW GPU tensor."""
    return (MODE != "miopen" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and not conv.weight.requires_grad
            and x.is_contiguous())


def report():
    """{"fwd": {"gemm": n, "miopen": m}, "bwd": {...}, "saved_ms": ...} over the shapes calibrated so far."""
    out = {"fwd": {"gemm": 0, "miopen": 0}, "bwd": {"gemm": 0, "miopen": 0}}
    saved = 0.0
    for key, algo in _choice.items():
        out[key[0]][algo] += 1
        g, m = _timings[key]
        if g < m:
            saved += m - g
    out["gemm_faster_by_ms_per_call_sum"] = round(saved, 3)
    return out


def reset():
    _choice.clear()
    _timings.clear()
