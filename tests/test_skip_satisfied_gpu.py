"""``DorPatch(skip_satisfied=True)`` (opt-in): the backward pass runs only over the EOT samples whose CW hinge is
active (``HotLoop._fb_taped`` + ``dorpatch_amd/taped.py``) — against the same step with every sample back-propagated
through autograd (``skip_satisfied=False``, what the reference does at ``attack.py:247``).  Skipping must change
nothing: a satisfied sample's logit gradient is exactly zero (``attack.py:16-23``), so its input gradient is too.

The confidence is chosen so that about half of the samples are satisfied; also covered: samples of one image split
over several micro-batches and whole images per micro-batch, backward batches padded up to a ladder size, an image
that has early-stopped, and a step in which nothing carries gradient.
tests/test_skip_satisfied_emu.py re-runs this module on CPU tensors through the host emulation, on a shallow ResNetV2."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dorpatch_amd.attack import DorPatch, HotLoop  # noqa: E402
from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, ResNetV2, seeded_init_  # noqa: E402
from dorpatch_amd.utils import NormModel, get_normalize  # noqa: E402

DEV = "cuda:0"
H, S, B = 224, 16, 2
LAYERS = (3, 4, 6, 3)
N_CLASSES = 1000
DETERMINISTIC = "auto"
LAYOUTS = {"split": dict(micro_batch=8, ladder=[2, 8]),         # each image's 16 samples over two micro-batches
           "whole": dict(micro_batch=16, ladder=[4, 16])}       # one image per micro-batch


class FixedDraw(object):
    def __init__(self, rows):
        self.rows = list(rows)

    def choice(self, a, n, replace=False):
        return np.asarray(self.rows.pop(0)).copy()


_cache = {}


def _problem():
    if "p" not in _cache:
        net = seeded_init_(ResNetV2(LAYERS, (256, 512, 1024, 2048), N_CLASSES), seed=1234,
                           gn_bias=WELL_CONDITIONED_GN_BIAS).fold_weight_standardization().freeze()
        model = NormModel(net, get_normalize("imagenet", "resnetv2")).eval()
        g = torch.Generator().manual_seed(5)
        x, mask, pattern = torch.rand(B, 3, H, H, generator=g), torch.rand(B, 1, H, H, generator=g), torch.rand(B, 3, H, H, generator=g)
        with torch.no_grad():
            y = model(x).topk(2)[1][:, 1].clone()                 # target = runner-up class
        rs = np.random.RandomState(5)
        idx = [rs.choice(2520 if H >= 224 else 630, S, replace=False) for _ in range(B)]
        _cache["p"] = (model.to(DEV), x, mask, pattern, y, idx)
    return _cache["p"]


def _step(confidence, skip, layout="split", stop_image=None, stage=0, **extras):
    model, x, mask, pattern, y, idx = _problem()
    cfg = LAYOUTS[layout]
    got = {}
    hook = lambda d: got.update({k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in d.items()})
    owner = DorPatch(micro_batch=cfg["micro_batch"], verbose=False, skip_satisfied=skip, deterministic=DETERMINISTIC)
    loop = HotLoop(owner, model, x.to(DEV), 0.12, N_CLASSES, "t/cfg/sub", 0, y.to(DEV), True, 1e-2, confidence, 0, 1, 10, 7,
                   'topk', 2, S, 1e-3, 1e-3, 4.0, False,
                   dict(init_mask=mask, init_pattern=pattern, rngs=[FixedDraw([i]) for i in idx], failure_refresh=10 ** 9,
                        step_hook=hook, tape_tabs=4, backward_ladder=cfg["ladder"], **extras))
    loop.stage = stage
    if stop_image is not None:
        loop.img[stop_image].active = False
    loop.step(1)
    if DEV != "cpu":
        torch.cuda.synchronize()
    loop.close()
    got["counts"] = (loop.n_forward, loop.n_active, loop.n_backward)
    got["taped"] = loop._taped
    return got


def _half_satisfied_confidence():
    if "conf" not in _cache:
        base = _step(0.1, skip=False)
        d = base["loss_adv"].reshape(-1) - 0.1                    # other - real per sample (every hinge active at 0.1)
        assert (base["loss_adv"] > 0).all()
        _cache["conf"] = float(-np.median(d))
    return _cache["conf"]


def _close(a, b, tol=3e-6):
    scale = float(b.abs().max())
    return float((a - b).abs().max()) <= tol * scale


@pytest.mark.parametrize("layout", ["split", "whole"])
def test_skipping_satisfied_samples_changes_nothing(layout):
    conf = _half_satisfied_confidence()
    ref = _step(conf, skip=False, layout=layout)
    got = _step(conf, skip=True, layout=layout)
    n_fwd, n_act, n_bwd = got["counts"]
    assert got["taped"] and not ref["taped"]
    assert n_fwd == B * S and 0 < n_act < n_fwd and n_act <= n_bwd < n_fwd          # some skipped, some padded
    assert n_act == int((ref["loss_adv"] > 0).sum())
    assert np.array_equal(got["loss_adv"], ref["loss_adv"])                          # the forward is the same kernels
    assert float(ref["g_adv"].abs().max()) > 0
    for name in ("g_adv", "grad_pattern", "grad_mask"):
        assert _close(got[name], ref[name]), name


def test_every_sample_active_and_no_sample_active():
    ref = _step(0.1, skip=False)
    got = _step(0.1, skip=True)
    assert got["counts"] == (B * S, B * S, B * S)
    assert np.array_equal(got["loss_adv"], ref["loss_adv"]) and _close(got["g_adv"], ref["g_adv"])
    none = _step(-1e4, skip=True)                                  # every margin met: nothing to back-propagate
    assert none["counts"] == (B * S, 0, 0) and not none["g_adv"].any() and not none["loss_adv"].any()


def test_early_stopped_image_is_not_back_propagated():
    got = _step(0.1, skip=True, stop_image=0, stage=1)
    ref = _step(0.1, skip=False, stage=1)
    assert got["counts"][1] == S and got["counts"][2] < B * S
    assert not got["g_adv"][0].any()                               # its update is lr = 0 anyway
    assert _close(got["g_adv"][1], ref["g_adv"][1])


def test_few_satisfied_samples_are_back_propagated_in_place():
    """Below ``skip_min_fraction`` (default 0.2) of skippable samples a group is not compacted: every sample goes through
    the backward in place (the zeros are computed, as in the reference); with the threshold at 0 the same step compacts."""
    base = _step(0.1, skip=False)
    d = np.sort(base["loss_adv"].reshape(-1) - 0.1)
    conf = float(-0.5 * (d[5] + d[6]))                             # exactly 6 of the 32 samples meet their margin (< 20 %)
    ref = _step(conf, skip=False)
    assert int((ref["loss_adv"] > 0).sum()) == B * S - 6
    got = _step(conf, skip=True)
    assert got["counts"] == (B * S, B * S - 6, B * S)
    tight = _step(conf, skip=True, skip_min_fraction=0.0)         # 26 active -> backward batches 8 + 8 + 8 + 2
    assert tight["counts"][1] == B * S - 6 and tight["counts"][2] == B * S - 6
    for run in (got, tight):
        assert np.array_equal(run["loss_adv"], ref["loss_adv"])
        for name in ("g_adv", "grad_pattern", "grad_mask"):
            assert _close(run[name], ref[name]), name
