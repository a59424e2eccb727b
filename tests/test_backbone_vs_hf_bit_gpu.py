"""The GPU backbone path — fused GroupNorm+ReLU(+residual), pad+max-pool, stem input gradient and subsample kernels, the
table-routed 1x1 convolutions, libconv — against an INDEPENDENT implementation evaluated in fp64: Hugging Face
transformers' ``BitForImageClassification`` (a port of timm's resnetv2.py that shares no code with this repository).

VERDICT r3 ("Backbone oracle is the same module"): tests/test_backbone_parity_gpu.py and smoke() compare the GPU with
dorpatch_amd/resnetv2.py itself run on the CPU in fp64, and the independent pin (tests/test_backbone_vs_hf_bit.py) ran
on the CPU eager path only.  Here the two meet: the same well-conditioned seeded weights are loaded into both networks
(timm key layout <-> transformers key layout), the product runs frozen + folded on the GPU in fp32, the reference
implementation on the CPU in fp64; logits and the input gradient must agree to fp32 round-off."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

transformers = pytest.importorskip("transformers")
try:
    from transformers import BitConfig, BitForImageClassification
except Exception as e:  # pragma: no cover
    pytest.skip("transformers has no BiT model: %r" % (e,), allow_module_level=True)

from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, resnetv2_50x1_bit, seeded_init_  # noqa: E402

DEV = "cuda:0"


def _to_hf_keys(state):
    """timm resnetv2 key layout (dorpatch_amd/resnetv2.py, the PatchCleanser checkpoint) -> transformers BiT."""
    out = {}
    for k, v in state.items():
        if k == "stem.conv.weight":
            out["bit.embedder.convolution.weight"] = v
        elif k.startswith("stages."):
            _, s, blocks, b, tail = k.split(".", 4)
            assert blocks == "blocks"
            out["bit.encoder.stages.%s.layers.%s.%s" % (s, b, tail)] = v
        elif k.startswith("norm."):
            out["bit.norm." + k[len("norm."):]] = v
        elif k == "head.fc.weight":
            out["classifier.1.weight"] = v.reshape(v.shape[0], v.shape[1])
        elif k == "head.fc.bias":
            out["classifier.1.bias"] = v
        else:
            raise KeyError(k)
    return out


@pytest.mark.parametrize("H,N", [(224, 3), (96, 2)])
def test_gpu_backbone_matches_transformers_bit_in_fp64(H, N):
    mine = seeded_init_(resnetv2_50x1_bit(1000), seed=1234, gn_bias=WELL_CONDITIONED_GN_BIAS).eval()
    cfg = BitConfig(num_labels=1000, layer_type="preactivation", global_padding=None, embedding_dynamic_padding=False,
                    num_groups=32, depths=[3, 4, 6, 3], hidden_sizes=[256, 512, 1024, 2048], embedding_size=64,
                    width_factor=1)
    hf = BitForImageClassification(cfg).eval()
    missing, unexpected = hf.load_state_dict(_to_hf_keys(mine.state_dict()), strict=True)
    assert not missing and not unexpected
    hf = hf.double()
    g = torch.Generator().manual_seed(17)
    x = torch.rand(N, 3, H, H, generator=g) * 2 - 1
    dl = torch.randn(N, 1000, generator=g)
    xa = x.double().requires_grad_(True)
    want = hf(pixel_values=xa).logits
    (gw,) = torch.autograd.grad(want, xa, dl.double())

    net = copy.deepcopy(mine).fold_weight_standardization().freeze().to(DEV)
    xb = x.to(DEV).requires_grad_(True)
    got = net(xb)
    (gg,) = torch.autograd.grad(got, xb, dl.to(DEV))
    scale = float(want.detach().abs().max())
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=0, atol=2e-5 * scale)
    a, b = gg.cpu().numpy().astype(np.float64), gw.numpy()
    rel = np.linalg.norm(a - b) / np.linalg.norm(b)
    err = np.abs(a - b) / np.abs(b).max()
    print("GPU vs transformers-BiT fp64 @%d: logits max err %.2e of scale; input gradient rel-L2 %.2e, max %.2e of scale"
          % (H, float(np.abs(got.detach().cpu().numpy() - want.detach().numpy()).max()) / scale, rel, err.max()))
    # the same bound smoke() states against the in-repo fp64 oracle (one flipped ReLU gate allowed)
    assert rel <= 5e-4 and (err > 1e-4).mean() <= 1e-3 and err.max() <= 5e-3, (rel, err.max(), (err > 1e-4).mean())
