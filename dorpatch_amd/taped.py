"""Explicit-tape forward and SELECTED-SAMPLE backward of the frozen ResNetV2 (the backbone of ``attack.py:222, 247``).

Why this exists.  The reference's adversarial loss is a hinge (``CW_loss``, ``attack.py:16-23``): an EOT sample whose
margin is already met contributes an exactly zero logit gradient, and because the backbone is frozen and normalises per
sample (GroupNorm), its whole input gradient is exactly zero too.  ``loss.sum().backward()`` (``attack.py:247``)
nevertheless pushes those zeros through all 50 layers.  Once an attack starts to succeed that is most of the backward
pass.  Autograd cannot back-propagate a subset of the rows of a recorded graph, so the hot loop records its own tape:

* ``forward`` runs one micro-batch through exactly the kernels of ``ResNetV2.forward`` (``dp_gn_relu_fwd``,
  ``dp_pad_maxpool_fwd``, the routed 1x1 convolutions, MIOpen 3x3) without autograd and appends what the backward
  needs — the GroupNorm inputs and statistics, the pooling codes — as one more "tab" of a ``StepTape``;
* ``backward`` takes any list of source samples (across all tabs of the step) and the matching logit-gradient rows and
  runs the input-gradient chain for those rows only: library backward-data convolutions on dense (M, ...) gradients,
  and ``dp_gn_relu_bwd_gather``, which reads each selected sample's saved activation in place (no gather copy).

With every sample selected the arithmetic is the autograd path's, call for call.  Nothing here knows about DorPatch's
losses; ``attack.HotLoop`` decides which samples carry gradient.
"""
import numpy as np
import torch

from . import conv1x1, libconv, ops
from .resnetv2 import GroupNormAct, ResNetV2, StdConv2d

MAX_TABS = 8          # kGnMaxTabs of dp_gn_relu_bwd_gather


class Unsupported(RuntimeError):
    """This network / input shape cannot take the taped path (the caller falls back to autograd)."""


def _need(ok, what):
    if not ok:
        raise Unsupported(what)


def eligible(net):
    """Static part of the check: dorpatch_amd's own ResNetV2, fully frozen, StdConv weights folded, fused kernels on."""
    if not (isinstance(net, ResNetV2) and GroupNormAct.fused):
        return False
    if any(p.requires_grad for p in net.parameters()):
        return False
    return all(m.folded for m in net.modules() if isinstance(m, StdConv2d))


class StepTape(object):
    """What the micro-batches ("tabs") of one step's forward leave behind for the backward.

    ``gn[j]`` (one entry per GroupNorm, in forward order) = ``{"norm", "x": [tab tensors], "mean", "rstd"}`` with the
    statistics of ALL tabs in one (capacity * tab_rows * groups,) buffer indexed by source sample = tab * tab_rows + row;
    every tab but the last holds exactly ``tab_rows`` samples."""

    def __init__(self, tab_rows, capacity):
        assert 1 <= capacity <= MAX_TABS
        self.tab_rows, self.capacity = int(tab_rows), int(capacity)
        self.rows = []            # samples per tab
        self.gn = []
        self.codes = []           # pooling codes per tab
        self.z_hw = None          # stem-conv output size
        self.conv_in = {}         # id(conv) -> (C, H, W) of its input
        self.conv_route = {}      # id(conv) -> True when the routed 1x1 path ran the forward
        self.dual = {}            # id(block) -> True when conv1 + downsample ran as the dual form
        self._code_cat = None

    @property
    def n_samples(self):
        return sum(self.rows)

    def nbytes(self):
        n = sum(t.numel() * t.element_size() for e in self.gn for t in e["x"])
        return n + sum(c.numel() for c in self.codes)

    def code_rows(self, sel):
        if self._code_cat is None:
            self._code_cat = self.codes[0] if len(self.codes) == 1 else torch.cat(self.codes)
        return self._code_cat if sel is None else self._code_cat.index_select(0, sel)


class _Cursor(object):
    def __init__(self, tape, n):
        self.tape, self.n, self.j = tape, n, 0
        self.k = len(tape.rows)

    def gn(self, norm, x, res, stats_only=False):
        tape, G = self.tape, norm.num_groups
        _need(ops.gn_relu_supported(x, G), "GroupNorm shape %s" % (tuple(x.shape),))
        if self.k == 0:
            cap = tape.capacity * tape.tab_rows * G
            tape.gn.append({"norm": norm, "x": [],
                            "mean": torch.empty((cap,), dtype=torch.float32, device=x.device),
                            "rstd": torch.empty((cap,), dtype=torch.float32, device=x.device)})
        e = tape.gn[self.j]
        self.j += 1
        lo = self.k * tape.tab_rows * G
        stats = (e["mean"][lo:lo + self.n * G], e["rstd"][lo:lo + self.n * G])
        if stats_only:       # the consuming convolution applies the norm while staging (round 5): keep x + statistics only
            _, _, ab, s = ops.gn_stats(x.contiguous(), norm.weight, norm.bias, G, norm.eps, stats_out=stats)
            e["x"].append(s)
            return ab, s
        y, _, _, s = ops.gn_relu_fwd(x.contiguous(), norm.weight, norm.bias, G, norm.eps,
                                     res=None if res is None else res.contiguous(), stats_out=stats)
        e["x"].append(s)
        return y, s


def _conv_fwd(tape, conv, x):
    """``StdConv2d.forward`` of a folded, frozen convolution, minus autograd."""
    x = x.contiguous()
    tape.conv_in[id(conv)] = tuple(x.shape[1:])
    routed = conv1x1.applicable(conv, x)
    tape.conv_route[id(conv)] = routed
    if routed:
        return conv1x1._run("fwd", x, conv.weight)
    return libconv.conv_fwd(x, conv.weight, conv.stride, conv.padding)


def _block_sum(tape, it, block, x):
    """``PreActBottleneck.forward_sum`` (the folded form: norms inside the consuming convolution's staging, the residual add
    in conv3's epilogue), kernel for kernel, minus autograd."""
    n1, n2, n3 = block.norm1, block.norm2, block.norm3
    c1, c2, c3 = block.conv1, block.conv2, block.conv3
    x = x.contiguous()
    fold1 = ops.gn_fold_supported(x, c1.weight, n1.num_groups)
    tape.conv_in[id(c1)] = tuple(x.shape[1:])
    if block.downsample is not None:
        ds = block.downsample.conv
        st = ds.stride[0]
        tape.dual[id(block)] = True
        if fold1 and ((x.shape[2] // st) * (x.shape[3] // st)) % 4 == 0:       # ops.GnDualConvFunction.forward
            ab, _ = it.gn(n1, x, None, stats_only=True)
            branch = ops._fconv_fwd(1, x, c1.weight, ab, None)
            small = x if st == 1 else ops.subsample2(x)
            shortcut = ops._fconv_fwd(1, small, ds.weight, ab, None)
            tape.conv_route[id(c1)] = tape.conv_route[id(ds)] = "mfma"
        else:                                                                   # ops.DualConv1x1Function.forward
            pre, _ = it.gn(n1, x, None)
            branch = conv1x1._run("fwd", pre, c1.weight)
            small = pre if st == 1 else ops.subsample2(pre)
            shortcut = conv1x1._run("fwd", small, ds.weight)
            tape.conv_route[id(c1)] = tape.conv_route[id(ds)] = True
        tape.conv_in[id(ds)] = tuple(small.shape[1:])
    elif fold1:
        ab, shortcut = it.gn(n1, x, None, stats_only=True)
        branch = ops._fconv_fwd(1, x, c1.weight, ab, None)
        tape.conv_route[id(c1)] = "mfma"
    else:
        pre, shortcut = it.gn(n1, x, None)
        branch = conv1x1._run("fwd", pre, c1.weight)
        tape.conv_route[id(c1)] = True
    if ops.gn_fold_supported(branch, c2.weight, n2.num_groups, c2.stride, c2.padding):
        ab, _ = it.gn(n2, branch, None, stats_only=True)
        kind = 3 if c2.stride[0] == 1 else 32
        tape.conv_in[id(c2)], tape.conv_route[id(c2)] = tuple(branch.shape[1:]), "mfma%d" % kind
        b = ops._fconv_fwd(kind, branch.contiguous(), c2.weight, ab, None)
    else:
        y2, _ = it.gn(n2, branch, None)
        b = _conv_fwd(tape, c2, y2)
    tape.conv_in[id(c3)], tape.conv_route[id(c3)] = tuple(b.shape[1:]), "mfma"
    if ops.gn_fold_supported(b, c3.weight, n3.num_groups):
        ab, _ = it.gn(n3, b, None, stats_only=True)
        return ops._fconv_fwd(1, b.contiguous(), c3.weight, ab, shortcut.contiguous())
    y3, _ = it.gn(n3, b, None)
    return ops._fconv_fwd(1, y3, c3.weight, None, shortcut.contiguous())


@torch.no_grad()
def forward(net, x, tape, z=None):
    """One micro-batch (n <= tape.tab_rows samples) -> logits; appends a tab to ``tape``.  ``z``: the stem convolution's
    output when the caller ran it (the stem-split form), else it is computed from ``x``."""
    k = len(tape.rows)
    _need(k < tape.capacity, "tape full")
    _need(k == 0 or tape.rows[-1] == tape.tab_rows, "only the last tab of a tape may be short")
    stem = net.stem.conv
    if z is None:
        z = libconv.conv_fwd(x, stem.weight, stem.stride, stem.padding)
    n = z.shape[0]
    _need(n <= tape.tab_rows, "micro-batch larger than the tape's tab")
    _need(ops.pad_maxpool_supported(z), "stem pooling shape %s" % (tuple(z.shape),))
    cur_it = _Cursor(tape, n)
    p, code = ops.pad_maxpool_fwd(z.contiguous())
    tape.z_hw = (z.shape[2], z.shape[3])
    cur, res = p, None
    for stage in net.stages:
        for block in stage.blocks:
            if res is None and block._sum_ok(cur):
                cur = _block_sum(tape, cur_it, block, cur)
                continue
            pre, s = cur_it.gn(block.norm1, cur, res)
            if block.downsample is not None:
                ds = block.downsample.conv
                dual = block._dual_ok(pre)
                tape.dual[id(block)] = dual
                if dual:          # ops.DualConv1x1Function.forward
                    branch = conv1x1._run("fwd", pre, block.conv1.weight)
                    small = pre if ds.stride[0] == 1 else ops.subsample2(pre)
                    shortcut = conv1x1._run("fwd", small, ds.weight)
                    tape.conv_in[id(block.conv1)] = tuple(pre.shape[1:])
                    tape.conv_in[id(ds)] = tuple(small.shape[1:])
                else:
                    shortcut = _conv_fwd(tape, ds, pre)
                    branch = _conv_fwd(tape, block.conv1, pre)
            else:
                shortcut = s
                branch = _conv_fwd(tape, block.conv1, pre)
            y2, _ = cur_it.gn(block.norm2, branch, None)
            b = _conv_fwd(tape, block.conv2, y2)
            y3, _ = cur_it.gn(block.norm3, b, None)
            cur, res = _conv_fwd(tape, block.conv3, y3), shortcut
    yf, _ = cur_it.gn(net.norm, cur, res)
    logits = net.head(yf)
    tape.codes.append(code)
    tape.rows.append(n)
    tape._code_cat = None
    return logits


class _Refs(object):
    """Dense never-touched tensors that only tell ATen's backward-data the input SHAPE (MIOpen never reads them)."""

    def __init__(self, like):
        self.like, self.cache = like, {}

    def get(self, shape):
        t = self.cache.get(shape)
        if t is None:
            t = self.cache[shape] = torch.empty(shape, dtype=self.like.dtype, device=self.like.device)
        return t


def _conv_bwd(tape, conv, dy, refs):
    """Input gradient of a frozen convolution for the M rows of ``dy``."""
    dy = dy.contiguous()
    route = tape.conv_route[id(conv)]
    if route == "mfma":                     # the folded form's convolutions: always the hand-written kernels
        return ops._fconv_bwd(1, dy, conv.weight)
    if route == "mfma3":
        return ops._fconv_bwd(3, dy, conv.weight)
    ref = refs.get((dy.shape[0],) + tape.conv_in[id(conv)])
    if route == "mfma32":                   # stride-2 3x3: own forward, the library's input gradient
        return ops._fconv_bwd(32, dy, conv.weight, x_ref=ref)
    if route:
        return conv1x1._run("bwd", dy, conv.weight, ref)
    return libconv.conv_bwd_data(dy, ref, conv.weight, conv.stride, conv.padding)


@torch.no_grad()
def backward(net, tape, dlogits, sel=None, through_stem=True):
    """d loss / d input for the selected source samples.

    ``sel``: (M,) int32 device tensor of source samples (tab * tab_rows + row), or None = all samples of a ONE-tab tape
    in order.  ``dlogits``: (M, n_classes) fp32.  Returns (M, 3, H, W), or with ``through_stem=False`` the gradient
    w.r.t. the stem convolution's output (M, 64, H/2, W/2)."""
    if sel is None:
        assert len(tape.rows) == 1
        sel = torch.arange(tape.rows[0], dtype=torch.int32, device=dlogits.device)
    M = int(sel.numel())
    assert dlogits.shape[0] == M
    refs = _Refs(dlogits)
    gn = list(tape.gn)

    def gn_bwd(dy, dres=None):
        e = gn.pop()
        norm = e["norm"]
        return ops.gn_relu_bwd_gather(dy.contiguous(), e["x"], tape.tab_rows, sel, norm.weight, norm.bias, e["mean"],
                                      e["rstd"], norm.num_groups, dres=None if dres is None else dres.contiguous())

    # head (resnetv2._Head): logits = fc(mean_hw(y)) -> d y = (dlogits @ W) / (H*W), the same for every pixel
    fc = net.head.fc
    C = fc.in_channels
    hf, wf = gn[-1]["x"][0].shape[2], gn[-1]["x"][0].shape[3]
    dpool = torch.mm(dlogits.to(fc.weight.dtype), fc.weight.view(fc.out_channels, C)) / float(hf * wf)
    dy = dpool.view(M, C, 1, 1).expand(M, C, hf, wf).contiguous()
    d_s = gn_bwd(dy)                                    # d loss / d (branch + shortcut) of the last block
    for stage in reversed(net.stages):
        for block in reversed(stage.blocks):
            d_branch = d_short = d_s
            d_b = gn_bwd(_conv_bwd(tape, block.conv3, d_branch, refs))
            d_a = gn_bwd(_conv_bwd(tape, block.conv2, d_b, refs))
            if block.downsample is not None:
                ds = block.downsample.conv
                if tape.dual[id(block)] and tape.conv_route.get(id(ds)) == "mfma":      # ops.GnDualConvFunction.backward
                    g = ops._fconv_bwd(1, d_a.contiguous(), block.conv1.weight)
                    if ds.stride[0] == 1:
                        g = ops._fconv_bwd(1, d_short.contiguous(), ds.weight, res=g)
                    else:
                        ops.subsample2_add_(g, ops._fconv_bwd(1, d_short.contiguous(), ds.weight))
                elif tape.dual[id(block)]:      # ops.DualConv1x1Function.backward
                    g = conv1x1._run("bwd", d_a.contiguous(), block.conv1.weight,
                                     refs.get((M,) + tape.conv_in[id(block.conv1)]))
                    if ds.stride[0] == 1:
                        O, Cin = ds.weight.shape[0], ds.weight.shape[1]
                        hw = d_short.shape[2] * d_short.shape[3]
                        g.view(M, Cin, hw).baddbmm_(ds.weight.view(O, Cin).t().unsqueeze(0).expand(M, Cin, O),
                                                    d_short.contiguous().view(M, O, hw))
                    else:
                        gs = conv1x1._run("bwd", d_short.contiguous(), ds.weight, refs.get((M,) + tape.conv_in[id(ds)]))
                        ops.subsample2_add_(g, gs)
                else:
                    g = _conv_bwd(tape, block.conv1, d_a, refs)
                    g = g + _conv_bwd(tape, ds, d_short, refs)
                d_s = gn_bwd(g)
            else:
                d_s = gn_bwd(_conv_bwd(tape, block.conv1, d_a, refs), dres=d_short)
    assert not gn
    dz = ops.pad_maxpool_bwd(d_s.contiguous(), tape.code_rows(sel).contiguous(), *tape.z_hw)
    if not through_stem:
        return dz
    return ops.stem_dgrad(dz, net.stem.conv.weight.contiguous())


# ---------------------------------------------------------------- which backward batches to run
def plan_chunks(n_active, ladder, cost):
    """Split ``n_active`` selected samples into backward batches whose sizes come from ``ladder`` (the batch sizes the
    library routes are tuned for; the last batch is padded up), minimising sum(cost[size]).  Returns [(real, size)]."""
    if n_active <= 0:
        return []
    ladder = sorted(ladder)
    unit = int(np.gcd.reduce(ladder))
    units = -(-n_active // unit)
    best = [0.0] + [float("inf")] * units
    pick = [0] * (units + 1)
    for r in range(1, units + 1):
        for L in ladder:
            c = cost[L] + best[max(0, r - L // unit)]
            if c < best[r] - 1e-12:
                best[r], pick[r] = c, L
    out, r, left = [], units, n_active
    while r > 0:
        L = pick[r]
        real = min(left, L)
        out.append((real, L))
        left -= real
        r = max(0, r - L // unit)
    out.sort(key=lambda t: -t[1])         # big batches first; the padded one (if any) is the last
    fixed, left = [], n_active
    for _, L in out:
        real = min(left, L)
        fixed.append((real, L))
        left -= real
    return [t for t in fixed if t[0] > 0]
