"""Whole-network parity of the hot-loop step through the real ResNetV2-50x1-BiT (reference
``attack.py:222, 247``: ``model(adv_x_masked)`` and ``loss.sum().backward()``), against the CPU oracle in
**fp64** (``oracle/restatement.eot_step`` on a ``.double()`` copy of the same weights).

Two weight sets (``dorpatch_amd/resnetv2.seeded_init_``):

* WELL-CONDITIONED (GroupNorm beta = 3.5): gate flips under re-ordered fp32 sums are rare, fp32 == fp64 to ~2e-6 of
  the gradient scale.  One ``HotLoop.step`` at 224x224, S = 8, both stages: every gradient within **1e-4 of the
  gradient scale** — with the one allowance a ReLU network needs: a single flipped gate moves a few hundred
  pixels of the input gradient by up to ~3e-4 of the scale (measured on the CPU: fp32 vs fp64, 1 of 3 seeds), so
  up to 0.1 % of the pixels may exceed the bound (and then the relative L2 error may reach a few 1e-4: measured
  on the GPU, stage 0: no flip, rel-L2 6e-7, every pixel within 7e-7 of the scale; stage 1: one flip, rel-L2
  1.2e-4, 0.027 % of the pixels beyond 1e-4, worst 1.1e-3).  A wrong constant, a mis-indexed group, a dropped
  residual or a bad tap anywhere in GroupNorm / stem / conv1x1 / pooling moves ALL pixels and fails the
  0.1 % bound.
* the benchmark's seeded-random weights (beta = 0): chaotic in fp32 (CPU fp32 vs fp64: 1.0-1.4e-2 rel-L2), so the
  statement is relative: the GPU's error against fp64 is no larger than 1.5x the CPU-fp32 error against fp64.

Also here: two fresh runs of the default configuration give BIT-IDENTICAL gradients (the
conv1x1 routing is a committed table, not a timing race).
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dorpatch_amd.attack import DorPatch, HotLoop  # noqa: E402
from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, resnetv2_50x1_bit, seeded_init_  # noqa: E402
from dorpatch_amd.utils import NormModel, get_normalize  # noqa: E402
from oracle import restatement as R  # noqa: E402

DEV = "cuda:0"
H, S = 224, 8


class FixedDraw(object):
    def __init__(self, rows):
        self.rows = list(rows)

    def choice(self, a, n, replace=False):
        return np.asarray(self.rows.pop(0)).copy()


def _problem(gn_bias, seed=1234, S=S):
    net = seeded_init_(resnetv2_50x1_bit(1000), seed=1234, gn_bias=gn_bias).fold_weight_standardization().freeze()
    model = NormModel(net, get_normalize("imagenet", "resnetv2")).eval()
    g = torch.Generator().manual_seed(seed)
    x, mask, pattern = torch.rand(1, 3, H, H, generator=g), torch.rand(1, 1, H, H, generator=g), torch.rand(1, 3, H, H, generator=g)
    with torch.no_grad():
        y = model(x).topk(2)[1][:, 1].clone()                 # target = runner-up class
    idx = np.random.RandomState(seed).choice(2520, S, replace=False)
    return model, x, mask, pattern, y, idx


def _oracle(model, x, mask, pattern, y, idx, stage, dtype):
    m = copy.deepcopy(model).to(dtype)
    keep = R.mask_universe(H, 2)[torch.from_numpy(idx)]
    return R.eot_step(m, x.to(dtype), mask.to(dtype), pattern.to(dtype), y, keep, stage=stage, targeted=True,
                      n_classes=1000, lr=0.01)


def _product(model, x, mask, pattern, y, idx, stage, **extras):
    S = len(idx)
    got = {}
    hook = lambda d: got.update({k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in d.items()})
    loop = HotLoop(DorPatch(verbose=False), copy.deepcopy(model).to(DEV), x.to(DEV), 0.12, 1000, "t/cfg/sub", 0,
                   y.to(DEV), True, 1e-2, 1e-1, 0, 1, 10, 7, 'topk', 2, S, 1e-3, 1e-3, 4.0, False,
                   dict(init_mask=mask, init_pattern=pattern, rngs=[FixedDraw([idx])], failure_refresh=10 ** 9,
                        step_hook=hook, **extras))
    loop.stage = stage
    loop.step(1)
    torch.cuda.synchronize()
    loop.close()
    return got


def _errors(got, want):
    a, b = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    scale = np.abs(b).max()
    e = np.abs(a - b) / scale
    return np.linalg.norm(a - b) / np.linalg.norm(b), float(e.max()), float((e > 1e-4).mean())


@pytest.mark.parametrize("stage", [0, 1])
def test_step_through_resnetv2_matches_fp64_oracle(stage):
    model, x, mask, pattern, y, idx = _problem(WELL_CONDITIONED_GN_BIAS)
    if stage == 1:
        mask = (mask > 0.9).float()
    want = _oracle(model, x, mask, pattern, y, idx, stage, torch.float64)
    got = _product(model, x, mask, pattern, y, idx, stage)
    assert (want["loss_adv"] > 0).all()                        # every sample's margin is active: all 8 carry gradient
    # the margin is a difference of two fp32 logits of magnitude ~5: a few 1e-5 absolute
    np.testing.assert_allclose(got["loss_adv"].reshape(-1), want["loss_adv"].numpy().reshape(-1), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(got["loss_struc"], want["loss_struc"].numpy(), rtol=2e-5)
    names = ["grad_pattern"] + (["grad_mask"] if stage == 0 else [])
    for name in names:
        rel, worst, frac = _errors(got[name].numpy(), want[name].numpy())
        print("stage %d %s: rel-L2 %.2e, max err / scale %.2e, pixels beyond 1e-4 of scale: %.2e" % (stage, name, rel, worst, frac))
        assert rel <= 5e-4 and frac <= 1e-3 and worst <= 5e-3, (name, rel, worst, frac)
    if stage == 0:
        np.testing.assert_allclose(got["group_lasso"], want["group_lasso"].numpy(), rtol=2e-5)
        np.testing.assert_allclose(got["density"], want["density"].numpy(), rtol=1e-4)


def test_gpu_fp32_is_as_close_to_fp64_as_cpu_fp32_on_the_chaotic_weights():
    """The benchmark's seeded-random weights: no fp32 evaluation can be held to 1e-4 (the reference's own CPU
    fp32 result misses fp64 by ~1e-2), so the GPU is held to the CPU's own miss, measured here, x 1.5."""
    model, x, mask, pattern, y, idx = _problem(0.0)
    w64 = _oracle(model, x, mask, pattern, y, idx, 0, torch.float64)
    w32 = _oracle(model, x, mask, pattern, y, idx, 0, torch.float32)
    got = _product(model, x, mask, pattern, y, idx, 0)
    err_cpu32 = _errors(w32["grad_pattern"].numpy(), w64["grad_pattern"].numpy())[0]
    err_gpu = _errors(got["grad_pattern"].numpy(), w64["grad_pattern"].numpy())[0]
    print("rel-L2 vs fp64: CPU fp32 %.3e, GPU fp32 %.3e" % (err_cpu32, err_gpu))
    assert 1e-4 < err_cpu32 < 1e-1                      # the premise (measured 1.0-1.4e-2 in the build container)
    assert err_gpu <= 1.5 * err_cpu32, (err_gpu, err_cpu32)
    np.testing.assert_allclose(got["loss_adv"].reshape(-1), w64["loss_adv"].numpy().reshape(-1), rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("n_masks", [8, 128])
def test_two_fresh_runs_are_bit_identical(n_masks):
    """Default configuration (conv1x1 table, MIOpen immediate mode, deterministic="auto" = per-problem policy): the same
    step twice from scratch gives the same bits — the optimiser takes sign(grad), so run-to-run reproducibility is part
    of parity.  8 masks: the batch at which MIOpen picks atomic split-K kernels for 8 convolution problems; 128: the
    reference's own problem size (1 image x sampling_size 128, attack.py:98), where exactly one problem needs forcing and
    the per-problem policy is what keeps the NHWC implicit-GEMM kernels for the rest."""
    from dorpatch_amd import conv1x1, libconv
    assert conv1x1.MODE == "table"
    model, x, mask, pattern, y, idx = _problem(0.0, S=n_masks)
    a = _product(model, x, mask, pattern, y, idx, 0)
    b = _product(model, x, mask, pattern, y, idx, 0)
    print("determinism policy after the runs: %s" % (libconv.summary(),))
    assert torch.equal(a["g_adv"], b["g_adv"]) and torch.equal(a["grad_pattern"], b["grad_pattern"])
    assert np.array_equal(a["loss_adv"], b["loss_adv"])


def test_fused_stem_reduction_is_bit_identical_to_the_autograd_path():
    """HotLoop(stem_split=True) for dorpatch_amd's own ResNetV2 — backward stops at the stem-conv output,
    dp_stem_dgrad_reduce produces the S-reduced patch gradient — against the default path (autograd down to the masked
    input + dp_apply_bwd): the same arithmetic in the same order, so the same bits."""
    model, x, mask, pattern, y, idx = _problem(0.0)
    a = _product(model, x, mask, pattern, y, idx, 0, stem_split=True)
    b = _product(model, x, mask, pattern, y, idx, 0)
    assert torch.equal(a["g_adv"], b["g_adv"]) and np.array_equal(a["loss_adv"], b["loss_adv"])
