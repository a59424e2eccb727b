"""Library (MIOpen) convolution calls of the frozen backbone, with PER-LAYER run-to-run determinism.

Why.  The optimiser takes ``sign(grad)`` (``attack.py:336-340``): two runs from identical seeds stay together only if
every kernel of the step is bit-reproducible.  The HIP kernels of this repo are (fixed-order reductions, no float
atomics); MIOpen's immediate mode, however, picks split-K implicit-GEMM kernels that accumulate with float atomics for
SOME (shape, batch) combinations — at 8 samples 5 of ResNetV2-50's 23 convolution shapes, forward and/or backward-data,
at 512 samples none (``profiles/r02c_determinism_probe.jsonl``).  ``torch.backends.cudnn.deterministic = True`` for the
whole run fixes that but bans MIOpen's whole NHWC implicit-GEMM family, also where it was deterministic anyway: 7.8 % at
the reference's own problem size (1 image x 128 masks: 29.98 vs 27.82 ms/step, ``profiles/r02w_final_tree_runs.json``).

So the decision is taken per (direction, batch, shape): the first time a convolution problem is seen under
``MODE == "auto"`` it is executed three times on its real operands; if any bit differs, that problem — and only that
problem — runs with deterministic library kernels from then on (the global flag is flipped around that one call).
The reference sets ``cudnn.benchmark = True`` (``utils.py:17``) and is not reproducible on a GPU at all.

``POLICY`` is process-global (a MIOpen property of the box, not of a run); ``merge_across(pg)`` ORs it over the ranks of
a process group so every replica forces the same problems.

**The stride-1 3x3 convolutions** (round 4).  MIOpen runs them as fp32 Winograd on the VALUs (108 TFLOP/s effective at
64 -> 64 @56^2, N = 512); ``dp_conv3x3_fwd`` — a direct implicit GEMM on the fp32 matrix cores, hand-written in
``csrc/dorpatch_hip.hip`` — runs the same convolution and, on transposed + flipped weights, its input gradient.  It is
exact f32 (an fmaf chain in a fixed order: deterministic by construction, so it never needs the probe above) and is used
for the (direction, channels, plane) problems the committed table ``conv3x3_gfx950.json`` lists as faster at the batch in
question (measured once on an MI355X by ``scripts/conv3x3_vs_miopen.py``; ``CONV3X3`` / the environment variable
``DORPATCH_CONV3X3`` = ``table`` (default) | ``on`` (every supported shape) | ``off`` (MIOpen)).  The packed weights
(one reshuffle per frozen filter and direction) are cached on the weight tensor itself.
"""
import contextlib
import json
import os

import torch
import torch.nn.functional as F

MODE = "off"        # "off": call the library as is;  "auto": probe each new problem once, force the non-reproducible ones
POLICY = {}         # (direction, N, C, O, k, stride, H, W) -> True (force deterministic kernels) | False (reproducible as is)
PROBE_RUNS = 3

CONV3X3 = os.environ.get("DORPATCH_CONV3X3", "table")
if CONV3X3 not in ("table", "on", "off"):
    raise ValueError("DORPATCH_CONV3X3 must be table, on or off, got %r" % CONV3X3)
_HERE = os.path.dirname(os.path.abspath(__file__))
try:
    with open(os.path.join(_HERE, "conv3x3_gfx950.json")) as _f:
        _doc3 = json.load(_f)
    CONV3X3_TABLE = {int(n): {tuple([k.split(":")[0]] + [int(v) for v in k.split(":")[1:]]): v for k, v in t.items()}
                     for n, t in _doc3["routes"].items()}
except (OSError, ValueError, KeyError) as _e:      # a missing / malformed table is a silent performance regression: say so
    import warnings
    warnings.warn("dorpatch_amd: conv3x3_gfx950.json unusable (%r): every 3x3 convolution stays on MIOpen" % (_e,))
    CONV3X3_TABLE = {}
for _extra in filter(None, os.environ.get("DORPATCH_CONV3X3_ALSO", "").split(",")):     # A/B knob: "fwd:512:7,bwd:512:7"
    _d, _c, _s = _extra.split(":")                                                       # -> also on dp_conv3x3_fwd, at
    CONV3X3_TABLE.setdefault(512, {})[(_d, int(_c), int(_s))] = "mfma"                   # batches >= 512
# The three stride-2 3x3 convolutions (round 5): forward on dp_conv3x3s2_fwd ("on", default: at every batch the kernel was
# measured at it beats MIOpen's NHWC implicit GEMM + transposes / stride-2 Winograd, profiles/r05h_*), or MIOpen ("off").
CONV3X3S2 = os.environ.get("DORPATCH_CONV3X3S2", "on")
if CONV3X3S2 not in ("on", "off"):
    raise ValueError("DORPATCH_CONV3X3S2 must be on or off, got %r" % CONV3X3S2)
# (round 6: with the 128- / 64-pixel tiles also at 32 rows — MIOpen / own 1.51, 1.30, 1.67 there, profiles/r06e_conv3x3s2_vs_miopen.jsonl)
CONV3X3S2_MIN_BATCH = int(os.environ.get("DORPATCH_CONV3X3S2_MIN_BATCH", "32"))
CONV3X3S2_BWD_MIN_BATCH = int(os.environ.get("DORPATCH_CONV3X3S2_BWD_MIN_BATCH", "64"))    # at 32 rows: 1.27, 1.20, 0.66
# Round 6: below ~192 workgroups (512 -> 512 @14 -> 7 at 64 samples: 7 tiles x 8 channel groups = 56) the kernel's 448-pixel tiles
# leave most of the chip idle (0.58 ms for 14.8 GFLOP, profiles/r06a_kernel_stats_timed_cfg3.txt) — but MIOpen has nothing
# better there either (its stride-2 Winograd: 0.63 ms, profiles/r06d_kernel_stats_timed_cfg3_fold.txt), so the threshold
# stays at 0 (DORPATCH_CONV3X3S2_MIN_WG is an A/B knob); the small-batch answer is a smaller tile, not the library.
CONV3X3S2_MIN_WG = int(os.environ.get("DORPATCH_CONV3X3S2_MIN_WG", "0"))


def conv3x3s2_fwd_pays(x, w):
    """dp_conv3x3s2_fwd's grid for this problem fills the chip (see CONV3X3S2_MIN_WG)."""
    so = int(x.shape[2]) // 2
    return (-(-int(x.shape[0]) * so * so // 448)) * (int(w.shape[0]) // 64) >= CONV3X3S2_MIN_WG
# ... and their input gradients on dp_conv3x3s2_bwd (masked parity-class walks over the dy plane, two column classes per
# workgroup) instead of MIOpen's NHWC implicit GEMM between batched_transpose_* kernels: "on" | "off".  Measured
# (profiles/r05k_*): MIOpen / own at N = 512: 1.32 (128 @28 -> 56), 1.08, 1.00; at N = 128: 1.21, 1.00, 0.96; at N = 64: 1.10,
# 1.01, 0.98 — and no transposes / fills around it; headline step 362.9 / 363.8 -> 361.0 / 361.3 ms on one box.
CONV3X3S2_BWD = os.environ.get("DORPATCH_CONV3X3S2_BWD", "on")
if CONV3X3S2_BWD not in ("on", "off"):
    raise ValueError("DORPATCH_CONV3X3S2_BWD must be on or off, got %r" % CONV3X3S2_BWD)
# The stem convolution (7x7 / stride 2, 3 -> 64 @224) on dp_stem_conv_fwd instead of MIOpen's stride-2 Winograd: "on" | "off".
STEM_CONV = os.environ.get("DORPATCH_STEM_CONV", "on")
if STEM_CONV not in ("on", "off"):
    raise ValueError("DORPATCH_STEM_CONV must be on or off, got %r" % STEM_CONV)
STEM_CONV_MIN_BATCH = int(os.environ.get("DORPATCH_STEM_CONV_MIN_BATCH", "16"))
_used3 = {}          # (direction, route) -> set of (N, C, S) routed (report_conv3x3())


def _conv3x3_route(direction, x, w, stride, padding):
    """True: run this stride-1 3x3 problem on dp_conv3x3_fwd.  ``x`` = the tensor the kernel would read (the input, or dy)."""
    if CONV3X3 == "off":
        return False
    from . import ops
    if not ops.conv3x3_supported(x, w if direction == "fwd" else w.transpose(0, 1), stride, padding):
        return False
    N, C, S = int(x.shape[0]), int(w.shape[1]), int(x.shape[2])
    if CONV3X3 == "on":
        use = True
    else:       # the column of the largest measured batch <= N; smaller batches than any measured: MIOpen
        cols = [n for n in sorted(CONV3X3_TABLE) if n <= N]
        use = bool(cols) and CONV3X3_TABLE[cols[-1]].get((direction, C, S)) == "mfma"
    _used3.setdefault((direction, "mfma" if use else "miopen"), set()).add((N, C, S))
    return use


def _packed(w, kind, make):
    """``make()`` cached on the weight tensor under ``kind`` (dies with the tensor; invalidated by an in-place update that
    bumps its version or by a move to another device).  Only FROZEN weights are routed here (``requires_grad`` False is
    checked by the callers): a write through ``w.data`` does not bump the version — call ``clear_packs(w)`` after one."""
    key = (w._version, w.data_ptr(), w.device)
    cache = getattr(w, "_dp_conv_pack", None)
    if cache is None or cache[0] != key:
        cache = (key, {})
        try:
            w._dp_conv_pack = cache
        except AttributeError:      # a tensor type that refuses attributes: pack every time
            pass
    if kind not in cache[1]:
        cache[1][kind] = make()
    return cache[1][kind]


def clear_packs(w):
    """Forget the packed copies of ``w`` (after editing a frozen weight through ``w.data``)."""
    if getattr(w, "_dp_conv_pack", None) is not None:
        try:
            del w._dp_conv_pack
        except AttributeError:
            pass


def _packed3(w, transpose):
    """pack_conv3x3_weights(w[, transposed + flipped]), cached."""
    from . import ops
    return _packed(w, ("3x3", bool(transpose)), lambda: ops.pack_conv3x3_weights(w, transpose=transpose))


def _packed3w(w, transpose):
    """pack_conv3x3_wino_weights(w[, transposed + flipped]), cached."""
    from . import ops
    return _packed(w, ("3x3-wino", bool(transpose)), lambda: ops.pack_conv3x3_wino_weights(w, transpose=transpose))


# Round 6: the stride-1 3x3 problems that run on own kernels take Winograd F(2x2, 3x3) on the matrix cores (dp_conv3x3_wino_fwd:
# 16 multiplications per 2 x 2 outputs instead of 36) where it is the faster one, else the direct kernels (dp_conv3x3_fwd):
# "on" | "off" (DORPATCH_CONV3X3_WINO).  Both are exact-f32, fixed-order, deterministic; they differ by fp32 round-off.
CONV3X3_WINO = os.environ.get("DORPATCH_CONV3X3_WINO", "on")
if CONV3X3_WINO not in ("on", "off"):
    raise ValueError("DORPATCH_CONV3X3_WINO must be on or off, got %r" % CONV3X3_WINO)
CONV3X3_WINO_MIN_BATCH = int(os.environ.get("DORPATCH_CONV3X3_WINO_MIN_BATCH", "32"))
_used_wino = set()


def conv3x3_s1(x, w, transpose, ab=None):
    """The stride-1 3x3 convolution of ``x`` with the frozen filter ``w`` (``transpose``: its input gradient on dy) on own
    kernels: Winograd where enabled and supported, else the direct implicit GEMM."""
    from . import ops
    if CONV3X3_WINO == "on" and x.shape[0] >= CONV3X3_WINO_MIN_BATCH and x.shape[2] in ops.CONV3X3_WINO_SIDES \
            and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
        _used_wino.add((int(x.shape[0]), int(w.shape[1]), int(x.shape[2]), bool(transpose)))
        return ops.conv3x3_wino_fwd(x, _packed3w(w, transpose), ab=ab)
    return ops.conv3x3_fwd(x, _packed3(w, transpose), ab=ab)


def _packed_stem(w):
    """pack_stem_weights(w), cached."""
    from . import ops
    return _packed(w, ("stem",), lambda: ops.pack_stem_weights(w))


def _packed3s2b(w):
    """pack_conv3x3s2_dgrad_weights(w), cached."""
    from . import ops
    return _packed(w, ("3x3s2-dgrad",), lambda: ops.pack_conv3x3s2_dgrad_weights(w))


def packed1(w, transpose):
    """pack_conv1x1_weights(w[, transposed]), cached."""
    from . import ops
    return _packed(w, ("1x1", bool(transpose)), lambda: ops.pack_conv1x1_weights(w, transpose=transpose))


def prepack(net):
    """Build every packed copy a frozen convolution of ``net`` can ask for, on the CURRENT stream, before the first
    forward.  ``_packed`` builds lazily on whatever stream is current; with micro-batches / sweep forwards round-robin on
    side streams (DorPatch(streams=2)) the first use would sit on side stream A and the second micro-batch, on stream B,
    would read the cached tensor with nothing ordering it after A's permute + copy.  Called by HotLoop and
    collect_failure (the two places that fork side streams; PatchCleanser's sweeps stay on one stream) before they do: those wait for the current stream, so every pack is
    complete for them.  Cheap when the packs exist (a dictionary lookup per filter)."""
    if not isinstance(net, torch.nn.Module):
        return 0
    n = 0
    for m in net.modules():
        w = getattr(m, "weight", None)
        if not (isinstance(m, torch.nn.Conv2d) and isinstance(w, torch.Tensor) and w.is_cuda and w.dtype == torch.float32
                and not w.requires_grad and getattr(m, "folded", False) and m.groups == 1):
            continue
        k, O, C = tuple(w.shape[2:]), int(w.shape[0]), int(w.shape[1])
        stride = tuple(m.stride)
        if k == (1, 1):
            if C % 16 == 0 and O % 64 == 0:
                packed1(w, False)
                n += 1
            if O % 16 == 0 and C % 64 == 0:
                packed1(w, True)
                n += 1
        elif k == (3, 3):
            if C % 8 == 0 and O % 64 == 0:
                _packed3(w, False)
                n += 1
            if stride == (1, 1) and O % 8 == 0 and C % 64 == 0:
                _packed3(w, True)
                n += 1
            if stride == (1, 1) and CONV3X3_WINO == "on" and O % 64 == 0 and C % 64 == 0:
                _packed3w(w, False)
                _packed3w(w, True)
                n += 2
            if stride == (2, 2) and O % 16 == 0 and C % 64 == 0:
                _packed3s2b(w)
                n += 1
        elif k == (7, 7) and (O, C) == (64, 3):
            _packed_stem(w)
            n += 1
    return n


def report_conv3x3():
    """{"mode", "fwd": {"mfma": n, "miopen": m}, "bwd": {...}}: distinct (batch, channels, plane) problems per route."""
    out = {"mode": CONV3X3}
    for d in ("fwd", "bwd"):
        out[d] = {r: len(_used3.get((d, r), ())) for r in ("mfma", "miopen")}
    return out


@contextlib.contextmanager
def _forced():
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        yield
    finally:
        torch.backends.cudnn.deterministic = was


def guard(key, fn, forced_fn=None):
    """Run ``fn()`` under the policy of problem ``key``.  ``forced_fn`` (default: ``fn`` itself under the deterministic
    flag) is what runs when the problem turned out not to be reproducible."""
    if MODE != "auto" or torch.backends.cudnn.deterministic:
        return fn()
    force = POLICY.get(key)
    if force is None:
        outs = [fn() for _ in range(PROBE_RUNS)]
        force = POLICY[key] = not all(torch.equal(outs[0], o) for o in outs[1:])      # host sync: once per problem
        if not force:
            return outs[-1]
    if not force:
        return fn()
    with _forced():
        return (forced_fn or fn)()


def _key(direction, n, w, stride, hw_in, padding=(0, 0)):
    return (direction, int(n), int(w.shape[1]), int(w.shape[0]), int(w.shape[2]), int(stride[0]), int(hw_in[0]),
            int(hw_in[1])) + ((int(padding[0]),) if int(padding[0]) != int(w.shape[2]) // 2 else ())   # "same" padding implied


def conv_fwd(x, w, stride=(1, 1), padding=(0, 0)):
    """``F.conv2d(x, w, None, stride, padding)`` for a frozen filter."""
    if w.shape[2] == 3 and _conv3x3_route("fwd", x, w, stride, padding):
        from . import ops
        return conv3x3_s1(x, w, False)
    if w.shape[2] == 3 and tuple(stride) == (2, 2) and CONV3X3S2 == "on" and x.shape[0] >= CONV3X3S2_MIN_BATCH:
        from . import ops
        if ops.conv3x3s2_supported(x, w, stride, padding) and conv3x3s2_fwd_pays(x, w):
            _used3.setdefault(("fwd", "mfma"), set()).add((int(x.shape[0]), int(w.shape[1]), -int(x.shape[2])))
            return ops.conv3x3s2_fwd(x, _packed3(w, False))
    if w.shape[2] == 7 and STEM_CONV == "on" and x.shape[0] >= STEM_CONV_MIN_BATCH:
        from . import ops
        if ops.stem_conv_supported(x, w, stride, padding):
            _used3.setdefault(("fwd", "mfma"), set()).add((int(x.shape[0]), 3, -7))
            return ops.stem_conv_fwd(x, _packed_stem(w))
    if MODE != "auto":
        return F.conv2d(x, w, None, stride, padding)
    return guard(_key("fwd", x.shape[0], w, stride, x.shape[2:], padding), lambda: F.conv2d(x, w, None, stride, padding))


def conv_bwd_data(dy, x_ref, w, stride=(1, 1), padding=(0, 0)):
    """Input gradient of the same convolution — exactly the call autograd makes (``x_ref`` is passed for its shape:
    MIOpen's backward-data never reads it, but ATen wants a dense tensor there)."""
    if w.shape[2] == 3 and dy.shape[2:] == x_ref.shape[2:] and _conv3x3_route("bwd", dy, w, stride, padding):
        from . import ops
        return conv3x3_s1(dy, w, True)
    if (w.shape[2] == 3 and tuple(stride) == (2, 2) and CONV3X3S2_BWD == "on" and dy.shape[0] >= CONV3X3S2_BWD_MIN_BATCH
            and x_ref.shape[2] == 2 * dy.shape[2] and x_ref.shape[3] == 2 * dy.shape[3]):
        from . import ops
        if ops.conv3x3s2_bwd_supported(dy, w, stride, padding):
            _used3.setdefault(("bwd", "mfma"), set()).add((int(dy.shape[0]), int(w.shape[1]), -int(dy.shape[2])))
            return ops.conv3x3s2_bwd(dy, _packed3s2b(w), int(w.shape[1]))

    def call():
        return torch.ops.aten.convolution_backward(dy, x_ref, w, None, tuple(stride), tuple(padding), (1, 1), False,
                                                   (0, 0), 1, (True, False, False))[0]
    if MODE != "auto":
        return call()
    return guard(_key("bwd", dy.shape[0], w, stride, x_ref.shape[2:], padding), call)


class FrozenConvFunction(torch.autograd.Function):
    """``conv2d`` with a frozen filter as an autograd node whose two library calls go through the policy above
    (plain autograd would make the same two calls, but outside this module's reach)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding):
        x = x.contiguous()
        ctx.save_for_backward(x, w)          # x: a shape reference for backward-data, never re-read
        ctx.geometry = (tuple(stride), tuple(padding))
        return conv_fwd(x, w, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding = ctx.geometry
        return conv_bwd_data(dy.contiguous(), x, w, stride, padding), None, None, None


def summary():
    """{"problems": n probed, "forced": m, "forced_list": [...]}: what ``deterministic="auto"`` decided so far."""
    forced = sorted(k for k, v in POLICY.items() if v)
    return {"problems": len(POLICY), "forced": len(forced), "forced_list": [list(k) for k in forced]}


def merge_across(pg):
    """OR the per-problem decisions over the ranks of ``pg`` (collective; call where every rank has seen the same
    problems — the all-samples autograd path — never from the rank-local selected-sample backward)."""
    if pg is None:
        return
    import torch.distributed as dist
    boxes = [None] * dist.get_world_size(pg)
    dist.all_gather_object(boxes, {k: v for k, v in POLICY.items()}, group=pg)
    for other in boxes:
        for k, v in other.items():
            POLICY[k] = bool(POLICY.get(k, False) or v)
