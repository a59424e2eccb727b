"""Per-problem determinism policy of the library convolutions (dorpatch_amd/libconv.py): a
convolution problem is probed the first time it is seen, only the non-reproducible ones are forced, the decisions are
process-global and OR-merged over ranks.  CPU: the policy logic with a deliberately non-reproducible callable, and the
frozen-convolution autograd node against plain autograd.  The GPU statement (two fresh runs of the whole network
bit-identical at the batch sizes where MIOpen does pick atomic kernels) is tests/test_backbone_parity_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dorpatch_amd import libconv


@pytest.fixture(autouse=True)
def _clean(monkeypatch):
    monkeypatch.setattr(libconv, "MODE", "auto")
    monkeypatch.setattr(libconv, "POLICY", {})
    yield


def test_probe_forces_only_the_problem_that_differs():
    calls = {"good": 0, "bad": 0, "bad_forced": 0}
    noise = iter(range(1, 100))

    def good():
        calls["good"] += 1
        return torch.ones(4)

    def bad():          # an atomics kernel: a different rounding every run ... unless the deterministic flag is on
        if torch.backends.cudnn.deterministic:
            calls["bad_forced"] += 1
            return torch.ones(4)
        calls["bad"] += 1
        return torch.ones(4) + 1e-6 * next(noise)

    was = torch.backends.cudnn.deterministic
    a = libconv.guard(("fwd", 8, 1, 1, 3, 1, 4, 4), good)
    assert calls["good"] == libconv.PROBE_RUNS and torch.equal(a, torch.ones(4))
    b = libconv.guard(("bwd", 8, 1, 1, 3, 2, 4, 4), bad)
    assert calls["bad"] == libconv.PROBE_RUNS and calls["bad_forced"] == 1 and torch.equal(b, torch.ones(4))
    assert torch.backends.cudnn.deterministic == was                      # the flag is flipped around the call only
    libconv.guard(("fwd", 8, 1, 1, 3, 1, 4, 4), good)
    libconv.guard(("bwd", 8, 1, 1, 3, 2, 4, 4), bad)
    assert calls == {"good": libconv.PROBE_RUNS + 1, "bad": libconv.PROBE_RUNS, "bad_forced": 2}   # no second probe
    s = libconv.summary()
    assert s["problems"] == 2 and s["forced"] == 1 and s["forced_list"] == [["bwd", 8, 1, 1, 3, 2, 4, 4]]
    # an alternative implementation for the forced case (the GEMM route falls back to MIOpen's deterministic kernel)
    out = libconv.guard(("gemm-fwd", 8, 1, 1, 1, 1, 4, 4), bad, forced_fn=lambda: torch.full((4,), 7.0))
    assert torch.equal(out, torch.full((4,), 7.0))


def test_off_mode_and_global_flag_bypass_the_probe(monkeypatch):
    n = []
    monkeypatch.setattr(libconv, "MODE", "off")
    libconv.guard(("fwd", 1, 1, 1, 1, 1, 1, 1), lambda: n.append(1) or torch.zeros(1))
    assert n == [1] and not libconv.POLICY
    monkeypatch.setattr(libconv, "MODE", "auto")
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    libconv.guard(("fwd", 1, 1, 1, 1, 1, 1, 1), lambda: n.append(1) or torch.zeros(1))
    assert n == [1, 1] and not libconv.POLICY


@pytest.mark.parametrize("k,stride,pad", [(3, 1, 1), (3, 2, 1), (7, 2, 3), (1, 2, 0)])
def test_frozen_conv_node_equals_autograd(k, stride, pad):
    g = torch.Generator().manual_seed(k * 10 + stride)
    x = torch.randn(3, 5, 12, 12, generator=g, requires_grad=True)
    w = torch.randn(6, 5, k, k, generator=g)
    y = libconv.FrozenConvFunction.apply(x, w, (stride, stride), (pad, pad))
    dy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, dy)
    x2 = x.detach().clone().requires_grad_(True)
    y2 = F.conv2d(x2, w, None, stride, pad)
    (gx2,) = torch.autograd.grad(y2, x2, dy)
    assert torch.equal(y, y2) and torch.equal(gx, gx2)
    keys = sorted(libconv.POLICY)
    assert keys == [("bwd", 3, 5, 6, k, stride, 12, 12), ("fwd", 3, 5, 6, k, stride, 12, 12)] and not any(libconv.POLICY.values())


def test_merge_across_ranks_is_an_or(monkeypatch):
    import torch.distributed as dist
    libconv.POLICY.update({("fwd", 8, 1, 1, 3, 1, 4, 4): False, ("bwd", 8, 1, 1, 3, 1, 4, 4): True})
    theirs = {("fwd", 8, 1, 1, 3, 1, 4, 4): True, ("fwd", 4, 1, 1, 3, 1, 4, 4): False}

    def fake_gather(boxes, mine, group=None):
        boxes[0], boxes[1] = dict(mine), theirs
    monkeypatch.setattr(dist, "get_world_size", lambda pg=None: 2)
    monkeypatch.setattr(dist, "all_gather_object", fake_gather)
    libconv.merge_across(object())
    assert libconv.POLICY == {("fwd", 8, 1, 1, 3, 1, 4, 4): True, ("bwd", 8, 1, 1, 3, 1, 4, 4): True,
                              ("fwd", 4, 1, 1, 3, 1, 4, 4): False}


def test_conv3x3_route_table_and_pack_cache(monkeypatch):
    """The stride-1 3x3 convolutions go to dp_conv3x3_fwd only for the (direction, channels, plane) problems the committed
    table lists at the largest measured batch <= N; CPU tensors never do (the kernel is GPU-only: no fallback is hidden
    here, the library call is the other ROUTE of the product).  The packed weights are cached on the weight tensor and
    rebuilt after an in-place update."""
    from dorpatch_amd import libconv, ops
    x, w = torch.randn(2, 64, 56, 56), torch.randn(64, 64, 3, 3)
    assert libconv._conv3x3_route("fwd", x, w, (1, 1), (1, 1)) is False                  # CPU tensor
    monkeypatch.setattr(ops, "conv3x3_supported", lambda *a, **k: True)
    monkeypatch.setattr(libconv, "CONV3X3_TABLE", {128: {("fwd", 64, 56): "mfma"}, 512: {("fwd", 64, 56): "miopen", ("bwd", 64, 56): "mfma"}})
    monkeypatch.setattr(libconv, "CONV3X3", "table")
    big, mid, small = torch.empty(512, 64, 56, 56, device="meta"), torch.empty(200, 64, 56, 56, device="meta"), torch.empty(8, 64, 56, 56, device="meta")
    assert libconv._conv3x3_route("fwd", mid, w, (1, 1), (1, 1)) is True                 # column 128
    assert libconv._conv3x3_route("fwd", big, w, (1, 1), (1, 1)) is False                # column 512 says miopen
    assert libconv._conv3x3_route("bwd", big, w, (1, 1), (1, 1)) is True
    assert libconv._conv3x3_route("fwd", small, w, (1, 1), (1, 1)) is False              # below every measured batch
    monkeypatch.setattr(libconv, "CONV3X3", "on")
    assert libconv._conv3x3_route("fwd", small, w, (1, 1), (1, 1)) is True
    monkeypatch.setattr(libconv, "CONV3X3", "off")
    assert libconv._conv3x3_route("fwd", mid, w, (1, 1), (1, 1)) is False
    a = libconv._packed3(w, False)
    assert libconv._packed3(w, False) is a and libconv._packed3(w, True) is not a
    assert a.shape == (1, 8, 4, 3, 3, 2, 64) and float(a[0, 2, 1, 0, 2, 1, 5]) == float(w[5, 2 * 8 + 2 + 1, 0, 2])
    t = libconv._packed3(w, True)                                                        # dgrad weights: transposed + flipped
    assert float(t[0, 2, 1, 0, 2, 1, 5]) == float(w[2 * 8 + 2 + 1, 5, 2, 0])
    w.mul_(2.0)
    b = libconv._packed3(w, False)
    assert b is not a and torch.equal(b, 2 * a)
    assert set(libconv.report_conv3x3()) == {"mode", "fwd", "bwd"}


def test_round5_weight_packings_and_routes():
    """The host-side packings of the round-5 kernels, element by element against the layouts include/dorpatch_hip.h states,
    and, for the stride-2 input gradient, against the convolution itself: summing each parity class's taps over the packed
    weights reproduces ATen's backward-data on a tiny problem (pure host arithmetic: no kernel involved)."""
    from dorpatch_amd import libconv, ops
    g = torch.Generator().manual_seed(5)
    O, C = 32, 64
    w = torch.randn(O, C, 3, 3, generator=g)
    # paired form: [row class 1 | row class 0], each [og][chunk][cp][th][j][half][c'], kw = (1, 2, 0)[j]
    p = ops.pack_conv3x3s2_dgrad_weights(w, pairs=True)
    assert p.numel() == 9 * O * C
    p1 = p[:6 * O * C].reshape(C // 64, O // 8, 4, 2, 3, 2, 64)
    p0 = p[6 * O * C:].reshape(C // 64, O // 8, 4, 1, 3, 2, 64)
    for (chunk, cp, th, j, half, c) in ((1, 2, 0, 0, 1, 5), (3, 0, 1, 2, 0, 63), (0, 3, 1, 1, 1, 17)):
        o = 8 * chunk + 2 * cp + half
        assert float(p1[0, chunk, cp, th, j, half, c]) == float(w[o, c, (2, 0)[th], (1, 2, 0)[j]])
        if th == 0:
            assert float(p0[0, chunk, cp, 0, j, half, c]) == float(w[o, c, 1, (1, 2, 0)[j]])
    # four-class form: classes (1,1), (0,1), (1,0), (0,0) with 16 / T channels per chunk
    q = ops.pack_conv3x3s2_dgrad_weights(w, pairs=False)
    q11 = q[:4 * O * C].reshape(C // 64, O // 4, 2, 2, 2, 2, 64)
    assert float(q11[0, 5, 1, 1, 0, 1, 9]) == float(w[4 * 5 + 2 + 1, 9, 0, 2])      # th = 1 -> kh 0, tw = 0 -> kw 2
    q00 = q[8 * O * C:].reshape(C // 64, O // 16, 8, 1, 1, 2, 64)
    assert float(q00[0, 1, 3, 0, 0, 0, 40]) == float(w[16 + 6, 40, 1, 1])
    # the classes ARE the input gradient: dx[2a + pr][2b + pc] = sum_{th <= pr, tw <= pc} dy[a + th][b + tw] w[kh][kw]
    N, S = 1, 3
    dy = torch.randn(N, O, S, S, generator=g)
    want = torch.ops.aten.convolution_backward(dy, torch.zeros(N, C, 2 * S, 2 * S), w, None, (2, 2), (1, 1), (1, 1), False,
                                               (0, 0), 1, (True, False, False))[0]
    dyp = F.pad(dy, (0, 1, 0, 1))
    got = torch.zeros_like(want)
    for pr in (0, 1):
        for pc in (0, 1):
            for th in range(pr + 1):
                for tw in range(pc + 1):
                    kh, kw = ((2, 0)[th] if pr else 1), ((2, 0)[tw] if pc else 1)
                    got[:, :, pr::2, pc::2] += torch.einsum("nohw,oc->nchw", dyp[:, :, th:th + S, tw:tw + S], w[:, :, kh, kw])
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    # stem: [7 j + kw][half][o] = w[o][r % 3][r // 3][kw], r = 2 j + half, the 22nd row zero
    ws = torch.randn(64, 3, 7, 7, generator=g)
    ps = ops.pack_stem_weights(ws).reshape(11, 7, 2, 64)
    for (j, kw, half, o) in ((0, 0, 0, 0), (4, 6, 1, 63), (10, 3, 0, 7)):
        r = 2 * j + half
        assert float(ps[j, kw, half, o]) == float(ws[o, r % 3, r // 3, kw])
    assert float(ps[10, :, 1].abs().max()) == 0.0
    # the committed route table knows the 384-input planes (round 5) and keeps the 224 ones
    assert libconv.CONV3X3_TABLE[64].get(("fwd", 64, 96)) == "mfma" and libconv.CONV3X3_TABLE[512].get(("bwd", 512, 7)) == "mfma"
    # round 6: with the Winograd kernel every 384-input shape is routed from 64 rows up (32 rows: 224-input shapes only)
    assert libconv.CONV3X3_TABLE[64].get(("fwd", 512, 12)) == "mfma" and libconv.CONV3X3_TABLE[32].get(("fwd", 512, 12)) is None
    assert libconv.CONV3X3S2_BWD in ("on", "off") and libconv.STEM_CONV in ("on", "off")
