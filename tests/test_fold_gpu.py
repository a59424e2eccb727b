"""Round 5 on the device: ResNetV2-50x1-BiT at 224 x 224, the folded graph (GroupNorm-apply + ReLU in the
consuming convolution's operand staging, residual adds in the producing convolution's epilogue) against the round-4 graph
with the same convolution kernels — logits bit-identical, input gradient equal to rounding — and at 384 x 384 (planes 96 /
48 / 24 / 12: the 3x3 convolutions stay on MIOpen there, the 1x1 ones fold)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from dorpatch_amd import conv1x1, ops, resnetv2  # noqa: E402

DEV = "cuda:0"


def _run(net, x, dl):
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        logits = net(xr)
    (g,) = torch.autograd.grad(logits, xr, dl)
    return logits.detach(), g


@pytest.mark.parametrize("N,side", [(4, 224), (64, 224), (2, 384)])
def test_folded_graph_equals_round4_graph_on_resnetv2_50(N, side, monkeypatch):
    from dorpatch_amd import libconv
    monkeypatch.setattr(conv1x1, "MODE", "mfma")
    monkeypatch.setattr(libconv, "CONV3X3", "on")       # the round-4 graph on the same 3x3 kernel whatever the batch size
    monkeypatch.setattr(resnetv2.GroupNormAct, "fold_min_batch", 1)
    monkeypatch.setattr(libconv, "CONV3X3S2_MIN_BATCH", 1)   # ... and on the same stride-2 forward kernel (dp_conv3x3s2_fwd)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)    # the stride-2 3x3 INPUT GRADIENTS stay on MIOpen (and
    # at 384 x 384 the forwards too): at small batches its default kernels accumulate with float atomics
    # (dorpatch_amd/libconv.py), which would differ run to run
    net = resnetv2.seeded_init_(resnetv2.resnetv2_50x1_bit(), gn_bias=resnetv2.WELL_CONDITIONED_GN_BIAS)
    net = net.fold_weight_standardization().freeze().to(DEV)
    gen = torch.Generator().manual_seed(N)
    x = torch.rand((N, 3, side, side), generator=gen).to(DEV)
    dl = torch.randn((N, 1000), generator=gen).to(DEV)
    calls = []
    orig = ops.GnConvFunction.forward

    def spy(ctx, *a):
        calls.append(a[6])
        return orig(ctx, *a)

    try:
        resnetv2.GroupNormAct.fold = False
        want, g_want = _run(net, x, dl)
        resnetv2.GroupNormAct.fold = True
        ops.GnConvFunction.forward = staticmethod(spy)
        got, g_got = _run(net, x, dl)
    finally:
        resnetv2.GroupNormAct.fold = True
        ops.GnConvFunction.forward = staticmethod(orig)
    assert calls.count(1) >= 20 and (side != 224 or (calls.count(3) >= 10 and calls.count(32) == 3))   # the folded nodes really ran
    assert torch.equal(got, want)
    scale = float(g_want.abs().max())
    assert scale > 0 and float((g_got - g_want).abs().max()) <= 2e-6 * scale
    got2, g_got2 = _run(net, x, dl)                                           # and reproducibly
    assert torch.equal(got2, got) and torch.equal(g_got2, g_got)
