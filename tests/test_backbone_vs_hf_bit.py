"""Pins the backbone restatement (dorpatch_amd/resnetv2.py) at the timm boundary against an INDEPENDENT
implementation of the same architecture that ships in this image: Hugging Face transformers'
``BitForImageClassification`` (models/bit/modeling_bit.py — transformers' port of timm's
``resnetv2.py``: weight-standardised convolutions eps = 1e-8, GroupNorm(32)+ReLU pre-activation
bottlenecks, stride on the 3x3, 'fixed' stem = conv7x7/2 -> ConstantPad2d(1) -> MaxPool 3/2).

timm==0.6.7 itself (reference requirements.txt:10, call site utils.py:51-63) is not installable
offline, and the reference pins no logits; this is the strongest anchor available here: the same
random weights, loaded through a key map into both networks, must give the same logits and the same
input gradient.  It also checks that timm's checkpoint key layout loads with strict=True.
"""
import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")
try:
    from transformers import BitConfig, BitForImageClassification
except Exception as e:  # pragma: no cover
    pytest.skip("transformers has no BiT model: %r" % (e,), allow_module_level=True)

from dorpatch_amd.resnetv2 import resnetv2_50x1_bit  # noqa: E402


def _hf_bit(num_labels=1000, seed=0):
    torch.manual_seed(seed)
    cfg = BitConfig(num_labels=num_labels, layer_type="preactivation", global_padding=None,
                    embedding_dynamic_padding=False, num_groups=32, depths=[3, 4, 6, 3],
                    hidden_sizes=[256, 512, 1024, 2048], embedding_size=64, width_factor=1)
    hf = BitForImageClassification(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():               # non-trivial affine parameters and classifier
        for name, p in hf.named_parameters():
            if ".norm" in name or name.startswith("bit.norm"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5 if name.endswith("weight")
                        else torch.randn(p.shape, generator=g) * 0.1)
            elif name.startswith("classifier"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return hf


def _to_timm_keys(hf_state):
    """transformers BiT state_dict -> timm resnetv2 key layout (the layout of the PatchCleanser checkpoint)."""
    out = {}
    for k, v in hf_state.items():
        if k == "bit.embedder.convolution.weight":
            out["stem.conv.weight"] = v
        elif k.startswith("bit.encoder.stages."):
            rest = k[len("bit.encoder.stages."):]
            s, layers, b, tail = rest.split(".", 3)
            assert layers == "layers"
            out["stages.%s.blocks.%s.%s" % (s, b, tail)] = v
        elif k.startswith("bit.norm."):
            out["norm." + k[len("bit.norm."):]] = v
        elif k == "classifier.1.weight":
            out["head.fc.weight"] = v.reshape(v.shape[0], v.shape[1], 1, 1)
        elif k == "classifier.1.bias":
            out["head.fc.bias"] = v
        else:
            raise KeyError(k)
    return out


@pytest.fixture(scope="module")
def pair():
    hf = _hf_bit()
    mine = resnetv2_50x1_bit(1000).eval()
    missing, unexpected = mine.load_state_dict(_to_timm_keys(hf.state_dict()), strict=True)
    assert not missing and not unexpected
    return hf, mine


def test_same_parameter_inventory(pair):
    hf, mine = pair
    assert sum(p.numel() for p in mine.parameters()) == sum(p.numel() for p in hf.parameters()) == 25549352
    assert len(mine.state_dict()) == 153                                  # 153 state tensors


@pytest.mark.parametrize("H", [224, 64])
def test_logits_and_input_gradient_match(pair, H):
    hf, mine = pair
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, H, H, generator=g) * 2 - 1
    dl = torch.randn(2, 1000, generator=g)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    want = hf(pixel_values=xa).logits
    got = mine(xb)
    scale = float(want.detach().abs().max())
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-5 * scale)
    (gw,) = torch.autograd.grad(want, xa, dl)
    (gg,) = torch.autograd.grad(got, xb, dl)
    a, b = gg.numpy().astype(np.float64), gw.numpy().astype(np.float64)
    # the two implementations standardise the weights by different formulas (batch_norm vs explicit): ulp-level
    # weight differences flip ReLU gates / max-pool argmaxes of this random-weight net, whose own fp32-vs-fp64
    # input-gradient spread is ~1e-2 rel-L2
    rel = np.linalg.norm(a - b) / np.linalg.norm(b)
    cos = (a * b).sum() / np.linalg.norm(a) / np.linalg.norm(b)
    assert rel < 3e-2 and cos > 0.999, (rel, cos)


def test_folded_weight_standardization_is_the_same_function(pair):
    """The hot path standardises the frozen weights once instead of every forward (StdConv2d.folded)."""
    import copy
    hf, mine = pair
    folded = copy.deepcopy(mine).fold_weight_standardization().freeze()
    x = torch.rand(1, 3, 96, 96, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        want, got = hf(pixel_values=x).logits, folded(x)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-5 * float(want.abs().max()))
