"""Differential tests against the UNMODIFIED reference executed in place (``oracle/ref_shim.py``) — only where
``/root/reference`` exists (the build container; skipped on the GPU box, which gets the recorded fixtures instead).

The committed fixtures pin fixed cases; here the same reference functions are swept over more inputs: mask geometry for
other image sizes / ratios / ``n_patch``, ``utils.clip`` and ``CW_loss`` on random tensors, ``patch_selection``, the
structural loss, and the helper functions ``main.py`` star-imports from ``utils``."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from dorpatch_amd import masks
from dorpatch_amd import utils as U
from oracle import ref_shim
from oracle import restatement as R

if not ref_shim.available():
    pytest.skip("reference tree not present (GPU box): the recorded fixtures cover this", allow_module_level=True)


@pytest.fixture(scope="module")
def ref():
    """The reference modules; the shim's process-wide patches (stub `torchvision` / `timm` modules, identity `.cuda()` on a
    GPU-less box) are undone when this module's tests are over."""
    import sys
    saved = (torch.Tensor.cuda, torch.nn.Module.cuda)
    mods_before = set(sys.modules)
    yield ref_shim.load_reference()
    torch.Tensor.cuda, torch.nn.Module.cuda = saved
    for name in set(sys.modules) - mods_before:
        if getattr(sys.modules[name], "__dorpatch_stub__", False):
            del sys.modules[name]


@pytest.mark.parametrize("H", [32, 56, 64, 98, 112])
@pytest.mark.parametrize("ratio", [0.015, 0.03, 0.06, 0.12, 0.2])
def test_mask_window_geometry_sweep(ref, H, ratio):
    """PatchCleanser.py:6-59 (n_patch = 1) vs the rectangle tables and vs the oracle's bool masks."""
    from dorpatch_amd.patchcleanser import MaskWindow
    with contextlib.redirect_stdout(io.StringIO()):
        mw = ref.PatchCleanser.MaskWindow(H, ratio, 1)
        mine = MaskWindow(H, ratio, 1, device="cpu")
    assert (mine.mask_size, mine.stride, mine.window_size) == (mw.mask_size, mw.stride, mw.window_size)
    assert torch.equal(mine.mask_set, mw.mask_set.bool()) and torch.equal(mine.double_mask_set, mw.double_mask_set.bool())
    assert torch.equal(mine.reverse_mask_set, mw.reverse_mask_set.bool())
    assert torch.equal(R.single_masks(H, ratio), mw.mask_set.bool())
    assert torch.equal(R.double_masks(H, ratio), mw.double_mask_set.bool())
    # and back: the bool form converts to an equivalent table (what collect_failure does with a bool universe)
    assert torch.equal(masks.rects_to_bool(masks.bool_to_rects(mw.double_mask_set.bool()), H), mine.double_mask_set)


@pytest.mark.parametrize("H,ratio", [(32, 0.06), (56, 0.03), (56, 0.12), (64, 0.2)])
def test_mask_window_two_patches(ref, H, ratio):
    """n_patch = 2 (PatchCleanser.py:31-38): mask_set = the 630 window pairs of the half-area windows, double_mask_set =
    36 x 630 triples, index i * 630 + k."""
    from dorpatch_amd.patchcleanser import MaskWindow
    with contextlib.redirect_stdout(io.StringIO()):
        mw = ref.PatchCleanser.MaskWindow(H, ratio, 2)
        mine = MaskWindow(H, ratio, 2, device="cpu")
    assert (mine.mask_size, mine.stride, mine.window_size) == (mw.mask_size, mw.stride, mw.window_size)
    assert torch.equal(mine.mask_set, mw.mask_set.bool())
    pick = torch.from_numpy(np.random.RandomState(0).choice(36 * 630, 400, replace=False))
    assert mine.double_rects.shape[0] == mw.double_mask_set.shape[0] == 36 * 630
    assert torch.equal(masks.rects_to_bool(mine.double_rects[pick.numpy()], H), mw.double_mask_set[pick].bool())


@pytest.mark.parametrize("dropout", [1, 2])
def test_get_mask_set_and_universe(ref, dropout):
    """attack.py:25-31, 83-85."""
    H = 56
    with contextlib.redirect_stdout(io.StringIO()):
        sets = [ref.attack.get_mask_set(H, r, dropout) for r in masks.DROPOUT_SIZES]
    uni = torch.cat(sets, 0).bool()
    assert torch.equal(masks.rects_to_bool(masks.universe_rects(H, dropout), H), uni)
    assert torch.equal(R.mask_universe(H, dropout), uni)


@pytest.mark.parametrize("eps", [0.5, 4.0, 1e4])
def test_clip_matches(ref, eps):
    """utils.py:105-110 incl. its (absent) gradient through the norm."""
    g = torch.Generator().manual_seed(int(eps * 10) + 1)
    m = torch.rand(2, 1, 28, 28, generator=g).requires_grad_(True)
    p = torch.rand(2, 3, 28, 28, generator=g).requires_grad_(True)
    x = torch.rand(2, 3, 28, 28, generator=g)
    m2, p2 = m.detach().clone().requires_grad_(True), p.detach().clone().requires_grad_(True)
    a, b = ref.utils.clip(m, p, x, eps), R.clip(m2, p2, x, eps)
    assert torch.equal(a, b)
    w = torch.rand(a.shape, generator=g)
    (a * w).sum().backward()
    (b * w).sum().backward()
    assert torch.equal(m.grad, m2.grad) and torch.equal(p.grad, p2.grad)


@pytest.mark.parametrize("targeted", [True, False])
@pytest.mark.parametrize("confidence", [0.0, 0.1, 5.0])
def test_cw_loss_matches(ref, targeted, confidence):
    """attack.py:10-23, both forms, incl. the gradient."""
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(12, 10, generator=g) * 3).requires_grad_(True)
    logits2 = logits.detach().clone().requires_grad_(True)
    y = torch.randint(0, 10, (12,), generator=g)
    a = ref.attack.CW_loss(10, targeted, confidence)(logits, y)
    b = R.cw_loss(logits2, y, 10, targeted, confidence)
    assert torch.equal(a, b)
    a.sum().backward()
    b.sum().backward()
    assert torch.equal(logits.grad, logits2.grad)


def test_structural_loss_and_its_odd_gradient(ref):
    """attack.py:33-45: gradient flows only through the subtracted neighbour."""
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 20, 24, generator=g).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    a, b = ref.attack.min_var_weighted_variance(x), R.min_var_weighted_variance(x2)
    assert torch.equal(a, b)
    a.sum().backward()
    b.sum().backward()
    assert torch.equal(x.grad, x2.grad)
    la, lb = ref.attack.local_variance(x.detach()), R.local_variance(x.detach())
    assert all(torch.equal(u, v) for u, v in zip(la, lb))


@pytest.mark.parametrize("budget", [0.0204, 0.06, 0.12])
def test_patch_selection_matches(ref, budget):
    """attack.py:363-382."""
    m = torch.rand(1, 1, 56, 56, generator=torch.Generator().manual_seed(int(budget * 1e4)))
    m[0, 0, :14] = 0                                      # cells with zero importance are never selected
    with contextlib.redirect_stdout(io.StringIO()):
        want = ref.attack.DorPatch().patch_selection(m.clone(), budget, 7, 'topk')
    assert torch.equal(R.patch_selection(m.clone(), budget, 7), want)


def test_utils_helpers_match(ref, tmp_path, monkeypatch):
    """What main.py star-imports (main.py:1): path mangling (utils.py:24-44), constants, float formatting."""
    monkeypatch.chdir(tmp_path)
    args = dict(device='0', dataset='imagenet', data_dir='/d', model_dir='m/', base_arch='resnetv2', targeted=True,
                patch_budget=0.06, attack='DorPatch', batch_size=1, epsilon=4., lr=0.01, num_patch=-1, dropout=2,
                density=1e-3, structured=1e-3)
    assert U.generate_saving_path(dict(args)) == ref.utils.generate_saving_path(dict(args))
    args.update(targeted=False, num_patch=4, patch_budget=0.12, structured=0.5)
    assert U.generate_saving_path(dict(args)) == ref.utils.generate_saving_path(dict(args))
    assert U.NUM_CLASSES_DICT == ref.utils.NUM_CLASSES_DICT
    for vals in ([1.0, 2.5], [0.015, 0.03, 0.06, 0.12], []):
        assert U.convert_float_list_to_str(vals) == ref.utils.convert_float_list_to_str(vals)
    a, b = U.get_normalize("imagenet", "resnetv2"), ref.utils.get_normalize("imagenet", "resnetv2")
    x = torch.rand(1, 3, 8, 8)
    assert torch.equal(a(x), b(x))
    net = torch.nn.Conv2d(3, 4, 1)
    assert torch.equal(U.NormModel(net, a)(x), ref.utils.NormModel(net, b)(x))
    assert os.path.isdir(U.generate_saving_path(dict(args)))


def test_patchcleanser_sweep_against_the_live_reference(ref):
    """defenses/PatchCleanser.py:68-112: the product's PatchCleanser (dp_apply_fwd + dp_argmax, here through the host
    emulation of the HIP kernels) against the unmodified reference's on more images than the recorded fixture holds (all
    three outcome kinds: first round disagrees / unanimous uncertified / unanimous certified), certify on and off —
    every record field identical."""
    from tests_hipemu import patch as emu_patch
    if emu_patch.build_emu.host_compiler() is None:
        pytest.skip("no host clang++ for the HIP emulation build")
    from dorpatch_amd.patchcleanser import MaskWindow, PatchCleanser
    from oracle import toy_models
    H = 56
    net = toy_models.NormModel(toy_models.make_peaky(), toy_models.Normalize())
    branches = set()
    with emu_patch.emulated_ops(), torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        for r in (0.03, 0.12):
            pc_ref = ref.PatchCleanser.PatchCleanser(ref.PatchCleanser.MaskWindow(H, r, 1), net)
            pc = PatchCleanser(MaskWindow(H, r, 1, device="cpu"), net)
            for seed in ((1, 4, 6, 9, 12, 19, 24, 28, 33, 36, 41, 47, 52) if r == 0.03 else (1, 8, 15, 20, 31, 44, 58)):
                img = toy_models.blob_image(H, seed)
                for certify in (True, False):
                    want, got = pc_ref.robust_predict(img, certify), pc.robust_predict(img, certify)
                    assert got.prediction == int(want.prediction) and bool(got.certification) == bool(want.certification), (r, seed)
                    assert np.array_equal(got.preds_1, want.preds_1), (r, seed)
                    assert (got.preds_2 is None) == (want.preds_2 is None)
                    if want.preds_2 is not None:
                        assert np.array_equal(got.preds_2, want.preds_2), (r, seed)
                    branches.add((len(np.unique(want.preds_1)) > 1, bool(want.certification)))
    assert len(branches) >= 3, branches


class _HalfSums(torch.nn.Module):
    """A classifier whose one-mask predictions TIE 18 : 18: class 3's logit is the sum over the left half of the image,
    class 7's the sum over the right half (everything else far below).  On an all-zero image the only non-zero pixels are
    the 0.5-filled occlusion window, so a first-round mask votes for the side it mostly covers — windows in columns 0-2 of
    the 6 x 6 grid say 3, columns 3-5 say 7."""

    def forward(self, x):
        half = x.shape[-1] // 2
        out = torch.full((x.shape[0], 10), -1e3, dtype=x.dtype, device=x.device)
        out[:, 3] = x[..., :half].sum((1, 2, 3))
        out[:, 7] = x[..., half:].sum((1, 2, 3))
        return out


def test_patchcleanser_majority_vote_on_a_count_tie(ref):
    """defenses/PatchCleanser.py:74-75: ``labels, counts = preds_1.unique(sorted=False, return_counts=True);
    label_majority = labels[counts.argmax()]`` — on a count tie the first of the tied labels IN THE ORDER ``unique``
    RETURNS THEM wins.  The product takes the smallest tied label (np.unique, ascending: what torch's GPU ``unique``
    returns whatever ``sorted`` says, and what this torch's CPU ``unique`` returns too — checked here, so a torch whose
    ``sorted=False`` order changes fails this test instead of silently changing the metric).  The recorded fixture holds
    no tie; this one is an exact 18 : 18 split, run through the unmodified reference and the product."""
    from tests_hipemu import patch as emu_patch
    if emu_patch.build_emu.host_compiler() is None:
        pytest.skip("no host clang++ for the HIP emulation build")
    from dorpatch_amd.patchcleanser import MaskWindow, PatchCleanser
    H, net = 56, _HalfSums()
    img = torch.zeros(3, H, H)
    with emu_patch.emulated_ops(), torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        for r in (0.03, 0.06):
            pc_ref = ref.PatchCleanser.PatchCleanser(ref.PatchCleanser.MaskWindow(H, r, 1), net)
            pc = PatchCleanser(MaskWindow(H, r, 1, device="cpu"), net)
            for certify in (True, False):
                want, got = pc_ref.robust_predict(img, certify), pc.robust_predict(img, certify)
                labels, counts = np.unique(want.preds_1, return_counts=True)
                assert labels.tolist() == [3, 7] and counts.tolist() == [18, 18], (r, labels, counts)     # the tie
                assert int(want.prediction) == 3                       # the reference resolves it to the smaller label
                assert got.prediction == int(want.prediction) and bool(got.certification) == bool(want.certification)
                assert np.array_equal(got.preds_1, want.preds_1)
                assert (got.preds_2 is None) == (want.preds_2 is None)
                if want.preds_2 is not None:
                    assert np.array_equal(got.preds_2, want.preds_2)
    l, c = torch.tensor([7, 3, 7, 3]).unique(sorted=False, return_counts=True)
    assert l.tolist() == [3, 7] and c.tolist() == [2, 2]               # the order the product's np.unique assumes


@pytest.mark.parametrize("hyper", [dict(eps=1.0, confidence=0.5, density=5e-3, structured=5e-3, patch_budget=0.06),
                                   dict(eps=8.0, confidence=0.0, density=1e-2, structured=1e-4, patch_budget=0.0204, lr=0.05)])
def test_hot_loop_steps_under_other_hyper_parameters(ref, hyper):
    """The reference's own `generate` run live with non-default eps / confidence / density / structured / patch_budget /
    lr (attack.py:51-53), its first steps of both stages recorded as in oracle/gen_golden.py, and replayed through the
    product's HotLoop.step (HIP kernels through the host emulation) and through the oracle — the recorded fixtures only
    hold the default hyper-parameters."""
    import importlib.util
    from tests_hipemu import patch as emu_patch
    if emu_patch.build_emu.host_compiler() is None:
        pytest.skip("no host clang++ for the HIP emulation build")
    from oracle import gen_golden as G
    H, S, n = 56, 6, 2
    net, x, y = G.toy_problem(H, gain=1.5, seed_x=11)
    kw = dict(hyper)
    eps = kw.pop("eps")
    budget = kw.pop("patch_budget")
    cap, _, _, _ = G.run_reference(net, x, y, sampling_size=S, max_iterations=n + 1, eps=eps, patch_budget=budget,
                                   keep=lambda s, i: True, **kw)
    g = G._pack_steps(cap, [(0, i) for i in range(n)] + [(1, i) for i in range(n)])
    g.update(x=x.numpy(), gain=1.5, H=H, S=S, eps=eps, budget=budget, confidence=hyper["confidence"], density=hyper["density"])
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("_attack_gpu_for_live", os.path.join(here, "test_attack_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.DEV = "cpu"
    saved = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with emu_patch.emulated_ops():
            mod._replay_golden_steps(g, 1e-3)
    finally:
        torch.cuda.synchronize = saved
    # and the oracle on the same recorded steps
    universe, lvx = R.mask_universe(H, 2), R.local_variance(x)[0].mean(1)
    for k in range(int(g["n_steps"])):
        p = "s%d_" % k
        out = R.eot_step(net, x, torch.from_numpy(g[p + "mask"]), torch.from_numpy(g[p + "pattern"]), torch.tensor([int(g[p + "y"])]),
                         universe[torch.from_numpy(g[p + "idx"])], stage=int(g[p + "stage"]), targeted=True, n_classes=10,
                         confidence=hyper["confidence"], density=hyper["density"], structured=float(g[p + "structured"]),
                         coeff_group_lasso=float(g[p + "coeff_group_lasso"]), eps=eps, local_var_x=lvx)
        np.testing.assert_allclose(out["loss_adv"].numpy().reshape(-1), g[p + "loss_adv"], atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(out["grad_pattern"].numpy(), g[p + "grad_pattern"], atol=1e-6, rtol=1e-4)
        if int(g[p + "stage"]) == 0:
            np.testing.assert_allclose(np.nan_to_num(out["grad_mask"].numpy()), np.nan_to_num(g[p + "grad_mask"]), atol=1e-6, rtol=1e-4)
