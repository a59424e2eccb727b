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
