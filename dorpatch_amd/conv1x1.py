"""Frozen 1x1 convolutions of the backbone: which LIBRARY kernel runs them.

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
Neither route is always faster (``profiles/r01d_conv1x1_table_n512.jsonl``).

``MODE`` (environment variable ``DORPATCH_CONV1X1`` at import):

``"table"`` (default)  the committed per-(gfx950, GEMM batch, direction, C, O, HW) table
                       ``conv1x1_gfx950.json``, measured once on an MI355X (two columns: with the libraries' default
                       GEMM solutions, and with the tuned solutions of ``tunableop_gfx950.csv`` — see
                       ``tuned_gemms_active``; both files are derived by ``scripts/make_conv1x1_table.py``);
                       shapes it does not list go to MIOpen.  Deterministic:
                       every process, every rank and every run executes the same kernels in the same
                       order, so two runs give bit-identical gradients (the optimiser takes ``sign(grad)``:
                       reproducibility outranks the last per cent).
``"auto"``             opt-in: time both routes at first use of each (direction, N, C, O, HW) and keep the
                       faster (MIOpen's find mode extended by one candidate).  A tie can flip between
                       runs, which changes the fp32 summation order (like the reference's
                       ``cudnn.benchmark = True``, ``utils.py:17``).  Under a process group call
                       ``share_choices(pg)`` after the warm-up so every rank runs rank 0's choices
                       (``HotLoop`` does).
``"gemm"`` / ``"miopen"``  force one route (``"miopen"`` = the plain ``F.conv2d`` path).

Weights are frozen on this path (``DorPatch.generate`` freezes the backbone): there is no weight
gradient.
"""
import json
import os
import time

import torch
import torch.nn.functional as F

MODES = ("table", "auto", "gemm", "miopen")
MODE = os.environ.get("DORPATCH_CONV1X1", "table")
if MODE not in MODES:
    raise ValueError("DORPATCH_CONV1X1 must be one of %s, got %r" % (", ".join(MODES), MODE))
CAL_ITERS = 3
_choice = {}          # auto mode: (direction, N, C, O, HW) -> "gemm" | "miopen"
_timings = {}         # same key -> (gemm_ms, miopen_ms)
_frozen = False       # auto mode: no further calibration (unknown shapes fall back to the table)
_used = {}            # (direction, route) -> number of distinct shapes routed (report())

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "conv1x1_gfx950.json")) as _f:
    _doc = json.load(_f)


def _parse(choices):
    return {tuple([k.split(":")[0]] + [int(v) for v in k.split(":")[1:]]): v for k, v in choices.items()}


PLAIN = {int(n): _parse(c) for n, c in _doc["plain"].items()}   # routes with the BLAS libraries' default GEMM solutions
TUNED = {int(n): _parse(c) for n, c in _doc["tuned"].items()}   # routes when the tuned GEMM solutions below are active
TUNED_BATCH = 512        # DorPatch's default micro-batch: the GEMM batch of the headline configuration
TABLE, TABLE_TUNED = PLAIN[TUNED_BATCH], TUNED[TUNED_BATCH]

# Tuned GEMM solutions for the GEMM route (PyTorch TunableOp, tuning done offline on an MI355X by
# scripts/tunableop_probe.py: 1.1-1.3x on most shapes, 2.9x on 64->64 and 1.9x on 256->64 @56x56, where hipBLASLt's
# default pick is poor).  Loaded once, at the first GPU call, with TUNING DISABLED: the file only maps GEMM shapes to
# library solution ids — still "MFMA left to rocBLAS/hipBLASLt", and still deterministic (no timing at run time).
# TunableOp validates the file against the installed PyTorch / HIP / rocBLAS / hipBLASLt versions and the GPU
# architecture and ignores it on a mismatch, in which case the plain table applies; before the first use a child
# process runs every tuned GEMM once (_selftest_tuned, ~10 s) and the file is only adopted if that process exits
# cleanly.  DORPATCH_TUNABLEOP=0 disables.
TUNABLEOP_FILE = os.path.join(_HERE, "tunableop_gfx950.csv")
TUNABLEOP = os.environ.get("DORPATCH_TUNABLEOP", "1") != "0"
_tuned_state = None      # None: not tried yet; True: solutions loaded; False: not in effect


_SELFTEST = r'''
import sys, torch
import torch.cuda.tunable as tun
sys.path.insert(0, sys.argv[2])
from dorpatch_amd import conv1x1
tun.enable(True); tun.tuning_enable(False); tun.record_untuned_enable(False)
if not tun.read_file(sys.argv[1]):
    sys.exit(3)
import os
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
dev = torch.device("cuda", torch.cuda.current_device())
for n, table in sorted(conv1x1.TUNED.items()):
    for (direction, C, O, HW), route in sorted(table.items()):
        if route != "gemm":
            continue
        H = int(round(HW ** 0.5))
        w = torch.randn(O, C, 1, 1, device=dev)
        t = torch.randn(n, C if direction == "fwd" else O, H, H, device=dev)
        out = conv1x1._IMPL[(direction, "gemm")](t, w, None)
        if not bool(torch.isfinite(out).all()):
            sys.exit(4)
torch.cuda.synchronize()
'''


def _selftest_tuned():
    """Run every GEMM of the tuned route columns once, with the tuned solutions, in a CHILD process: a solution id
    that a particular box's BLAS build rejects (or that faults) then costs a disabled feature, not the run.  The
    verdict is passed to child processes through DORPATCH_TUNABLEOP_VERIFIED."""
    verdict = os.environ.get("DORPATCH_TUNABLEOP_VERIFIED")
    if verdict in ("0", "1"):
        return verdict == "1"
    import subprocess
    import sys
    env = dict(os.environ, DORPATCH_TUNABLEOP="0")          # the child loads the file explicitly
    try:
        rc = subprocess.run([sys.executable, "-c", _SELFTEST, TUNABLEOP_FILE, os.path.dirname(_HERE)], env=env,
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180).returncode
    except (OSError, subprocess.SubprocessError):
        rc = -1
    os.environ["DORPATCH_TUNABLEOP_VERIFIED"] = "1" if rc == 0 else "0"
    return rc == 0


def tuned_gemms_active(device_is_cuda=True):
    """Load the tuned-solution file on first use (GPU only); -> whether it is in effect."""
    global _tuned_state
    if _tuned_state is None and device_is_cuda:
        _tuned_state = False
        if TUNABLEOP and os.path.exists(TUNABLEOP_FILE) and _selftest_tuned():
            try:
                import torch.cuda.tunable as tun
                tun.enable(True)
                tun.tuning_enable(False)
                tun.record_untuned_enable(False)
                _tuned_state = bool(tun.read_file(TUNABLEOP_FILE))
                if not _tuned_state:
                    tun.enable(False)
            except Exception:            # noqa: BLE001 - any refusal simply leaves the default solutions in place
                _tuned_state = False
    return bool(_tuned_state)


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


def _from_table(direction, C, O, HW, cuda=False, N=None):
    """Route of one (GEMM batch, direction, shape).  The tuned column applies only to batch sizes the solution file was
    tuned for (other batches run the libraries' default solutions even with the file loaded); batches without their own
    plain column use the 512-sample one; shapes nobody measured go to MIOpen."""
    key = (direction, C, O, HW)
    if N in TUNED and tuned_gemms_active(cuda):
        return TUNED[N].get(key, "miopen")
    return PLAIN.get(N, PLAIN[TUNED_BATCH]).get(key, "miopen")


def _pick(direction, t, w4d, x):
    O, C = w4d.shape[0], w4d.shape[1]
    HW = t.shape[2] * t.shape[3]
    if MODE in ("gemm", "miopen"):
        algo = MODE
    elif MODE == "table":
        algo = _from_table(direction, C, O, HW, t.is_cuda, t.shape[0])
    else:
        key = (direction, t.shape[0], C, O, HW)
        algo = _choice.get(key)
        if algo is None and _frozen:
            algo = _from_table(direction, C, O, HW, t.is_cuda, t.shape[0])
        elif algo is None:
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
    _used.setdefault((direction, t.shape[0], C, O, HW), algo)
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


def share_choices(pg):
    """auto mode under a process group: every rank adopts rank 0's calibrated choices and stops calibrating,
    so all replicas run the same kernels (shapes rank 0 has not seen fall back to the table)."""
    global _frozen
    if MODE != "auto" or pg is None:
        return
    from . import dist as dp_dist
    mine = dict(_choice)
    theirs = dp_dist.broadcast_object(mine, pg)
    _choice.clear()
    _choice.update(theirs)
    _frozen = True


def report_tuned():
    return "tuned GEMM solutions active (tunableop_gfx950.csv)" if _tuned_state else "library default GEMM solutions"


def report():
    """{"fwd": {"gemm": n, "miopen": m}, "bwd": {...}} over the distinct shapes routed so far (+ what the
    calibration measured, in auto mode)."""
    out = {"fwd": {"gemm": 0, "miopen": 0}, "bwd": {"gemm": 0, "miopen": 0}}
    for key, algo in _used.items():
        out[key[0]][algo] += 1
    if MODE == "auto":
        saved = 0.0
        for key in _choice:
            g, m = _timings.get(key, (0.0, 0.0))
            if g < m:
                saved += m - g
        out["gemm_faster_by_ms_per_call_sum"] = round(saved, 3)
    return out


def reset():
    global _frozen
    _choice.clear()
    _timings.clear()
    _used.clear()
    _frozen = False
