"""Round 5: the folded graph of the frozen ResNetV2 — GroupNorm-apply + ReLU inside the consuming
convolution's operand staging (ops.GnConvFunction / GnDualConvFunction over dp_gn_stats + dp_conv1x1_fwd), the residual add
in the producing convolution's epilogue — against the round-4 graph (one GroupNorm kernel per norm, adds fused into the next
norm) with the SAME convolution kernels: logits must be bit-identical, the input gradient equal to rounding (the stride-1
downsample gradient accumulates in the kernel's epilogue instead of a GEMM with beta = 1).  HIP kernels, including the
matrix-core convolutions, through the host emulation on CPU tensors (HIPEMU_MFMA_CONVS=1)."""
import os

import pytest
import torch

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

from dorpatch_amd import conv1x1, resnetv2  # noqa: E402


def _net(layers=(2,), n_classes=10):
    """ONE stage (two bottlenecks, 64 -> 256 channels) by default: a 448-pixel x 64-channel MFMA tile costs the fibre
    emulation seconds whatever the plane size, so the narrow net keeps this file at CPU-suite scale (DORPATCH_EMU_FULL=1 adds
    a stride-2 stage); the full 50-layer network takes the same code on the GPU (tests/test_fold_gpu.py)."""
    if os.environ.get("DORPATCH_EMU_FULL", "0") == "1" and layers == (2,):
        layers = (2, 1)
    net = resnetv2.ResNetV2(layers, (256, 512, 1024, 2048)[:len(layers)], n_classes)
    resnetv2.seeded_init_(net, 7, gn_bias=resnetv2.WELL_CONDITIONED_GN_BIAS)
    net.fold_weight_standardization()
    return net.freeze()


def _run(net, x, dl):
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        logits = net(xr)
    (g,) = torch.autograd.grad(logits, xr, dl)
    return logits.detach(), g


@pytest.fixture
def mfma_convs(monkeypatch):
    monkeypatch.setenv("HIPEMU_MFMA_CONVS", "1")
    monkeypatch.setattr(conv1x1, "MODE", "mfma")
    monkeypatch.setattr(resnetv2.GroupNormAct, "fold_min_batch", 1)
    yield


def test_folded_graph_equals_round4_graph(mfma_convs):
    net = _net()
    gen = torch.Generator().manual_seed(3)
    x = torch.rand((2, 3, 32, 32), generator=gen)        # planes 8x8 (/ 4x4)
    dl = torch.randn((2, 10), generator=gen)
    used = []
    from dorpatch_amd import ops
    orig = ops.GnConvFunction.forward

    def spy(ctx, *a):
        used.append((tuple(a[0].shape), a[6], a[7] is not None, a[8]))
        return orig(ctx, *a)

    with emu_patch.emulated_ops():
        try:
            resnetv2.GroupNormAct.fold = False
            want, g_want = _run(net, x, dl)
            resnetv2.GroupNormAct.fold = True
            ops.GnConvFunction.forward = staticmethod(spy)
            got, g_got = _run(net, x, dl)
        finally:
            resnetv2.GroupNormAct.fold = os.environ.get("DORPATCH_GNFOLD", "1") != "0"
            ops.GnConvFunction.forward = staticmethod(orig)
    # stage 0 (2 blocks @8x8) and stage 1 (@4x4) took the folded form: conv3 with the epilogue add, the second block's conv1
    # with the pass-through shortcut
    assert ((2, 64, 8, 8), 1, True, False) in used and ((2, 256, 8, 8), 1, False, True) in used
    assert torch.equal(got, want)
    scale = float(g_want.abs().max())
    assert scale > 0 and float((g_got - g_want).abs().max()) <= 2e-6 * scale


def test_folded_graph_is_off_when_the_library_routes_are_forced(mfma_convs, monkeypatch):
    """conv1x1.MODE = gemm / miopen (A/B switches) must bypass the hand-written kernels altogether."""
    monkeypatch.setattr(conv1x1, "MODE", "gemm")
    net = _net((1,))
    x = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(4))
    with emu_patch.emulated_ops():
        blk = net.stages[0].blocks[0]
        z = net.stem.pool(net.stem._conv(x))
        assert not blk._sum_ok(z)
