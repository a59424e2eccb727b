"""dorpatch_amd/taped.py — the explicit-tape forward and the selected-sample backward of the frozen ResNetV2-50x1-BiT —
against autograd through the same network, with the HIP kernels (incl. dp_gn_relu_bwd_gather) running under the host
emulation on CPU tensors.  Selecting every sample must reproduce the autograd gradient; selecting a subset (in any
order, across micro-batch "tabs", with a short last tab) must reproduce exactly those rows."""
import numpy as np
import pytest
import torch

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

from dorpatch_amd import resnetv2, taped  # noqa: E402


def _net(n_classes=10, layers=(2, 1, 1, 2)):
    """A shallow ResNetV2 of the same block types (first blocks with stride-1 / stride-2 downsample, plain blocks whose
    shortcut gradient is added inside the GroupNorm backward); the whole 50-layer network runs the same code through
    HotLoop in tests/test_attack_emu-style tests and on the GPU."""
    torch.manual_seed(0)
    net = resnetv2.ResNetV2(layers, (256, 512, 1024, 2048), n_classes)
    resnetv2.seeded_init_(net, 7, gn_bias=resnetv2.WELL_CONDITIONED_GN_BIAS)
    net.fold_weight_standardization()
    return net.freeze()


def _autograd(net, x, dl):
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        logits = net(xr)
    (g,) = torch.autograd.grad(logits, xr, dl)
    return logits.detach(), g


def test_all_samples_selected_equals_autograd(size=32):
    net = _net()
    gen = torch.Generator().manual_seed(1)
    x = torch.rand((2, 3, size, size), generator=gen)
    dl = torch.randn((2, 10), generator=gen)
    with emu_patch.emulated_ops():
        assert taped.eligible(net)
        logits, g = _autograd(net, x, dl)
        tape = taped.StepTape(tab_rows=2, capacity=1)
        logits_t = taped.forward(net, x, tape)
        g_t = taped.backward(net, tape, dl)
    assert torch.equal(logits, logits_t)
    scale = float(g.abs().max())
    assert scale > 0 and float((g - g_t).abs().max()) <= 2e-6 * scale
    assert tape.nbytes() > 0 and tape.n_samples == 2


def test_subset_across_tabs_with_short_last_tab_and_stem_split():
    net = _net()
    gen = torch.Generator().manual_seed(2)
    x = torch.rand((3, 3, 32, 32), generator=gen)
    dl = torch.randn((3, 10), generator=gen)
    with emu_patch.emulated_ops():
        ref_logits, g = _autograd(net, x, dl)
        tape = taped.StepTape(tab_rows=2, capacity=3)            # tabs of 2 and 1 samples
        logits = torch.cat([taped.forward(net, x[0:2], tape), taped.forward(net, x[2:3], tape)])
        sel = torch.tensor([2, 0], dtype=torch.int32)
        g_sel = taped.backward(net, tape, dl[sel.long()], sel)
        # padding rows (zero logit gradient) give exactly zero input gradient: what makes skipping them exact;
        # stopping at the stem convolution's output + dp_stem_dgrad is the same arithmetic
        pad = torch.tensor([2, 0, 0], dtype=torch.int32)
        dl_pad = dl[pad.long()].clone()
        dl_pad[2:] = 0
        from dorpatch_amd import ops
        dz = taped.backward(net, tape, dl_pad, pad, through_stem=False)
        g_pad = ops.stem_dgrad(dz, net.stem.conv.weight.contiguous())
        with pytest.raises(taped.Unsupported):
            taped.forward(net, x[0:1], tape)                     # a short tab must stay the last one
    # (the host convolutions are not batch-invariant to the last bit: 2 + 2 + 1 samples vs 5 at once)
    assert float((logits - ref_logits).abs().max()) <= 1e-5 * float(ref_logits.abs().max())
    scale = float(g.abs().max())
    assert float((g_sel - g[sel.long()]).abs().max()) <= 2e-6 * scale
    assert float((g_pad[:2] - g_sel).abs().max()) <= 2e-6 * scale and not g_pad[2:].any()


def test_ineligible_networks_are_refused():
    net = _net()
    assert taped.eligible(net)
    net.norm.weight.requires_grad_(True)
    assert not taped.eligible(net)
    assert not taped.eligible(torch.nn.Linear(3, 3))


def test_plan_chunks_prefers_cheap_covers():
    cost = {32: 40.0, 64: 70.0, 512: 512.0}
    assert taped.plan_chunks(0, [32, 64, 512], cost) == []
    assert taped.plan_chunks(512, [32, 64, 512], cost) == [(512, 512)]
    assert taped.plan_chunks(500, [32, 64, 512], cost) == [(500, 512)]          # padding 12 beats nine small batches
    assert taped.plan_chunks(70, [32, 64, 512], cost) == [(64, 64), (6, 32)]
    assert taped.plan_chunks(20, [32, 64, 512], cost) == [(20, 32)]
    plan = taped.plan_chunks(1300, [32, 64, 512], cost)
    assert sum(r for r, _ in plan) == 1300 and all(r <= s for r, s in plan) and [s for _, s in plan][:2] == [512, 512]
    assert taped.plan_chunks(7, [4], {4: 4.0}) == [(4, 4), (3, 4)]
