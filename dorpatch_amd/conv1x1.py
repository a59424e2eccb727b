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

MODES = ("table", "auto", "gemm", "miopen", "mfma")
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


# Round 5: shapes that run on the hand-written fp32-MFMA kernel dp_conv1x1_fwd instead (measured per (batch, direction, shape)
# against the faster library route by scripts/conv1x1_vs_lib.py; scripts/make_conv1x1_mfma_table.py).  Applies to the 1x1
# convolutions that are NOT part of a folded GroupNorm -> convolution node (those always run on the kernel: ops.GnConvFunction).
try:
    with open(os.path.join(_HERE, "conv1x1_mfma_gfx950.json")) as _f:
        MFMA = {int(n): set(_parse(c)) for n, c in json.load(_f)["routes"].items()}
except (OSError, ValueError, KeyError) as _e:
    import warnings
    warnings.warn("dorpatch_amd: conv1x1_mfma_gfx950.json unusable (%r): the un-folded 1x1 convolutions stay on the libraries" % (_e,))
    MFMA = {}

PLAIN = {int(n): _parse(c) for n, c in _doc["plain"].items()}   # routes with the BLAS libraries' default GEMM solutions
TUNED = {int(n): _parse(c) for n, c in _doc["tuned"].items()}   # routes when the tuned GEMM solutions below are active
TUNED_BATCH = 512        # DorPatch's default micro-batch: the GEMM batch of the headline configuration
TABLE, TABLE_TUNED = PLAIN[TUNED_BATCH], TUNED[TUNED_BATCH]

# Tuned GEMM solutions for the GEMM route (PyTorch TunableOp, tuning done offline on an MI355X by
# scripts/tunableop_probe.py: 1.1-1.3x on most shapes, 2.9x on 64->64 and 1.9x on 256->64 @56x56, where hipBLASLt's
# default pick is poor).  The file only maps GEMM shapes to library solution ids — still "MFMA left to
# rocBLAS/hipBLASLt", and still deterministic (TUNING DISABLED: no timing at run time).
#
# Life cycle (one verdict per process tree, one scope per generate()):
#   activate(pg)   called by HotLoop.__init__: the first call decides the VERDICT — rank 0 runs every tuned GEMM once in a
#                  child process, with the tuned solution and with the library default, and compares the two results
#                  numerically (_selftest_tuned: max |tuned - default| <= 1e-4 of the result's scale per shape; a
#                  solution id that faults costs the child, not the run); the verdict is broadcast, then every rank loads
#                  the file (TunableOp validates it against the installed PyTorch / HIP / rocBLAS / hipBLASLt versions and
#                  the GPU architecture) and the ranks agree on the AND of their results — so all replicas run the same
#                  kernels.  It then switches TunableOp on (tuning off) and remembers the state it found.
#   deactivate()   called by HotLoop.close(): TunableOp back to the state the caller had — GEMMs of the user's process
#                  outside generate() (PatchCleanser, their own model) never see the file.
# Between the two, batch sizes that have a tuned route column use it; outside, the plain columns and the libraries' default
# solutions apply.  DORPATCH_TUNABLEOP=0 disables the whole mechanism.
TUNABLEOP_FILE = os.path.join(_HERE, "tunableop_gfx950.csv")
TUNABLEOP = os.environ.get("DORPATCH_TUNABLEOP", "1") != "0"
SELFTEST_RTOL = 1e-4     # per tuned GEMM: max |tuned - default| / max |default|
_tuned_verdict = None    # None: undecided; True: file verified + loaded in this process; False: not usable here
_tuned_scope = 0         # > 0: inside activate() ... deactivate()
_tuned_prev = None       # TunableOp's (enabled, tuning) before the outermost activate()
_selftest_report = None  # what the self-test measured (bench.py / last_run record it)


_SELFTEST = r"""
import json, os, sys, torch
import torch.cuda.tunable as tun
sys.path.insert(0, sys.argv[2])
from dorpatch_amd import conv1x1
rtol = float(sys.argv[3])
tun.enable(True); tun.tuning_enable(False); tun.record_untuned_enable(False)
if not tun.read_file(sys.argv[1]):
    print(json.dumps({"ok": False, "why": "TunableOp rejected the file (validators)"})); sys.exit(3)
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
dev = torch.device("cuda", torch.cuda.current_device())
gen = torch.Generator(device="cpu").manual_seed(0)
worst, n, bad = 0.0, 0, []
for batch, table in sorted(conv1x1.TUNED.items()):
    for (direction, C, O, HW), route in sorted(table.items()):
        if route != "gemm":
            continue
        H = int(round(HW ** 0.5))
        w = (torch.randn(O, C, 1, 1, generator=gen) / C ** 0.5).to(dev)
        t = torch.randn(batch, C if direction == "fwd" else O, H, H, generator=gen).to(dev)
        tun.enable(True)
        got = conv1x1._IMPL[(direction, "gemm")](t, w, None)
        tun.enable(False)
        want = conv1x1._IMPL[(direction, "gemm")](t, w, None)
        err = float((got - want).abs().max() / want.abs().max())
        n += 1
        if not (err <= rtol):          # NaN fails too
            bad.append([batch, direction, C, O, HW, err])
        worst = max(worst, err) if err == err else float("inf")
torch.cuda.synchronize()
print(json.dumps({"ok": not bad, "gemms": n, "max_rel_err": worst, "rtol": rtol, "bad": bad}))
sys.exit(0 if not bad else 4)
"""


def _selftest_tuned():
    """Every GEMM of the tuned route columns, tuned solution vs library default, in a CHILD process (see above).
    The verdict reaches child processes of this one through DORPATCH_TUNABLEOP_VERIFIED."""
    global _selftest_report
    verdict = os.environ.get("DORPATCH_TUNABLEOP_VERIFIED")
    if verdict in ("0", "1"):
        _selftest_report = _selftest_report or {"ok": verdict == "1", "inherited": True}
        return verdict == "1"
    import subprocess
    import sys
    env = dict(os.environ, DORPATCH_TUNABLEOP="0")          # the child loads the file explicitly
    report = {"ok": False}
    try:
        res = subprocess.run([sys.executable, "-c", _SELFTEST, TUNABLEOP_FILE, os.path.dirname(_HERE), repr(SELFTEST_RTOL)],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, text=True)
        lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
        if lines:
            report = json.loads(lines[-1])
        report["returncode"] = res.returncode
        ok = res.returncode == 0 and bool(report.get("ok"))
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        ok = False
        report = {"ok": False, "why": repr(e)}
    _selftest_report = report
    os.environ["DORPATCH_TUNABLEOP_VERIFIED"] = "1" if ok else "0"
    return ok


def _load_file():
    """TunableOp reads (and validates) the solution file in this process; its enabled / tuning state is left as found."""
    try:
        import torch.cuda.tunable as tun
        was = (tun.is_enabled(), tun.tuning_is_enabled())
        tun.enable(True)
        tun.tuning_enable(False)
        tun.record_untuned_enable(False)
        ok = bool(tun.read_file(TUNABLEOP_FILE))
        tun.enable(was[0])
        tun.tuning_enable(was[1])
        return ok
    except Exception:            # noqa: BLE001 - any refusal simply leaves the default solutions in place
        return False


def _local_verdict(selftest):
    """This process's own answer, no collective: the file exists, (``selftest``) its solutions pass the numeric self-test
    in a child process, and TunableOp accepts it here."""
    if not (TUNABLEOP and os.path.exists(TUNABLEOP_FILE)):
        return False
    if selftest and not _selftest_tuned():
        return False
    return _load_file()


def _agree(pg):
    """The ranks of ``pg`` settle on ONE verdict — always the same two collectives (a broadcast of rank 0's answer, an
    all-gather of every rank's), whatever each rank has cached: a rank that already decided locally (e.g. it ran
    PatchCleanser before its first generate()) and a rank that has not must not enter different collectives.
    Rank 0's answer includes the numeric self-test; one rank's refusal is every rank's."""
    global _tuned_verdict
    from . import dist as dp_dist
    _, rank = dp_dist.world_rank(pg)
    mine = _tuned_verdict
    if mine is None:
        mine = _local_verdict(selftest=(rank == 0))
    lead = bool(dp_dist.broadcast_object(bool(mine), pg))
    _tuned_verdict = dp_dist.all_true(bool(mine) and lead, pg)


def activate(pg=None, device_is_cuda=True):
    """Enter a tuned-solution scope (see the life-cycle note above); -> whether the tuned solutions are in effect.
    With a process group every call is collective (``_agree``); without one the verdict is this process's own and is
    cached — a later activate(pg) still runs the agreement."""
    global _tuned_verdict, _tuned_scope, _tuned_prev
    if not device_is_cuda:
        return False
    if pg is not None:
        # every activate(pg) runs the two (cheap) collectives: a cache keyed on the group object's address could hit on some
        # ranks and miss on others after a group was destroyed and another created at the same address — a collective mismatch
        _agree(pg)
    elif _tuned_verdict is None:
        _tuned_verdict = _local_verdict(selftest=True)
    if not _tuned_verdict:
        return False
    import torch.cuda.tunable as tun
    if _tuned_scope == 0:
        _tuned_prev = (tun.is_enabled(), tun.tuning_is_enabled())
        tun.enable(True)
        tun.tuning_enable(False)
    _tuned_scope += 1
    return True


def deactivate():
    """Leave the scope opened by the matching ``activate`` that returned True."""
    global _tuned_scope, _tuned_prev
    if _tuned_scope == 0:
        return
    _tuned_scope -= 1
    if _tuned_scope == 0 and _tuned_prev is not None:
        import torch.cuda.tunable as tun
        tun.enable(_tuned_prev[0])
        tun.tuning_enable(_tuned_prev[1])
        _tuned_prev = None


def tuned_gemms_active(device_is_cuda=True):
    """Are the tuned solutions (and therefore the tuned route columns) in effect right now?  Pure query."""
    return bool(device_is_cuda and _tuned_scope > 0 and _tuned_verdict)


def _fwd_gemm(x, w4d, _=None):
    N, C, H, W = x.shape
    O = w4d.shape[0]
    return torch.bmm(w4d.view(1, O, C).expand(N, O, C), x.view(N, C, H * W)).view(N, O, H, W)


def _fwd_miopen(x, w4d, _=None):
    from . import libconv
    return libconv.conv_fwd(x, w4d)


def _bwd_gemm(dy, w4d, x=None):
    N, O, H, W = dy.shape
    C = w4d.shape[1]
    return torch.bmm(w4d.view(O, C).t().unsqueeze(0).expand(N, C, O), dy.view(N, O, H * W)).view(N, C, H, W)


def _bwd_miopen(dy, w4d, x):
    # exactly the call autograd makes for F.conv2d (input gradient only); `x` is passed for its shape —
    # MIOpen's backward-data never reads it, but ATen wants a dense tensor there
    from . import libconv
    return libconv.conv_bwd_data(dy, x, w4d)


def _fwd_mfma(x, w4d, _=None):
    from . import libconv, ops
    return ops.conv1x1_fwd(x, libconv.packed1(w4d, False))


def _bwd_mfma(dy, w4d, x=None):
    from . import libconv, ops
    return ops.conv1x1_fwd(dy, libconv.packed1(w4d, True))


_IMPL = {("fwd", "gemm"): _fwd_gemm, ("fwd", "miopen"): _fwd_miopen, ("fwd", "mfma"): _fwd_mfma,
         ("bwd", "gemm"): _bwd_gemm, ("bwd", "miopen"): _bwd_miopen, ("bwd", "mfma"): _bwd_mfma}


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
    if cuda and N is not None:
        cols = [n for n in MFMA if n <= N]
        if cols and key in MFMA[max(cols)]:
            return "mfma"
    if N in TUNED and tuned_gemms_active(cuda):
        return TUNED[N].get(key, "miopen")
    return PLAIN.get(N, PLAIN[TUNED_BATCH]).get(key, "miopen")


def _pick(direction, t, w4d, x):
    O, C = w4d.shape[0], w4d.shape[1]
    HW = t.shape[2] * t.shape[3]
    if MODE in ("gemm", "miopen"):
        algo = MODE
    elif MODE == "mfma":        # every shape dp_conv1x1_fwd takes on the hand-written kernel, the rest as the table says
        from . import ops
        algo = "mfma" if ops.conv1x1_supported(t, w4d if direction == "fwd" else w4d.transpose(0, 1)) else \
            _from_table(direction, C, O, HW, t.is_cuda, t.shape[0])
    elif MODE == "table":
        algo = _from_table(direction, C, O, HW, t.is_cuda, t.shape[0])
        if algo == "mfma":      # the table knows shapes, not sizes: a micro-batch beyond the kernel's 32-bit offsets takes
            from . import ops   # the library route of the same table instead of failing in DP_REQUIRE
            if not ops.conv1x1_supported(t, w4d if direction == "fwd" else w4d.transpose(0, 1)):
                algo = _from_table(direction, C, O, HW, False, t.shape[0])
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
    algo = _pick(direction, t, w4d, x)
    if algo == "gemm":
        from . import libconv
        if libconv.MODE == "auto":
            # a BLAS solution that accumulates with atomics (split-K / stream-K) would not be reproducible either:
            # probed like the MIOpen problems; the forced alternative is MIOpen's deterministic kernel
            key = ("gemm-" + direction, int(t.shape[0]), int(w4d.shape[1]), int(w4d.shape[0]), 1, 1, int(t.shape[2]),
                   int(t.shape[3]))
            if direction == "bwd" and x is None:
                x = torch.empty((t.shape[0], w4d.shape[1]) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            return libconv.guard(key, lambda: _IMPL[(direction, "gemm")](t, w4d, x),
                                 lambda: _IMPL[(direction, "miopen")](t, w4d, x))
    return _IMPL[(direction, algo)](t, w4d, x)


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
    """What generate() runs with in this process (the verdict of ``activate``), for bench.py / last_run."""
    return "tuned GEMM solutions active (tunableop_gfx950.csv)" if _tuned_verdict else "library default GEMM solutions"


def selftest_report():
    return _selftest_report


def report():
    """{"fwd": {"gemm": n, "miopen": m}, "bwd": {...}} over the distinct shapes routed so far (+ what the
    calibration measured, in auto mode)."""
    out = {"fwd": {"gemm": 0, "miopen": 0, "mfma": 0}, "bwd": {"gemm": 0, "miopen": 0, "mfma": 0}}
    for key, algo in _used.items():
        out[key[0]][algo] = out[key[0]].get(algo, 0) + 1
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
