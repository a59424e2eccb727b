"""PatchCleanser: the oracle restatement and the product's rectangle
tables against records of the UNMODIFIED reference (tests/golden/patchcleanser_56.npz)."""
import numpy as np
import torch

from dorpatch_amd import masks
from dorpatch_amd.patchcleanser import MaskWindow, PatchCleanserRecord, certified_metrics
from oracle import restatement as R
from oracle import toy_models


def _cases(g):
    return [(int(s), float(r)) for s, r in g["cases"]]


def test_restatement_reproduces_reference_records(golden_patchcleanser):
    g = golden_patchcleanser
    H = int(g["H"])
    net = toy_models.NormModel(toy_models.make_peaky(), toy_models.Normalize())
    kinds = set()
    for k, (seed, r) in enumerate(_cases(g)):
        img = toy_models.blob_image(H, seed)
        with torch.no_grad():
            np.testing.assert_allclose(net(img[None]).numpy(), g["c%d_logits" % k], rtol=1e-5, atol=1e-6)
        pred, cert, p1, p2 = R.patchcleanser_predict(net, img, R.single_masks(H, r), R.double_masks(H, r), True)
        assert pred == int(g["c%d_pred" % k]) and cert == bool(g["c%d_cert" % k]), (seed, r)
        assert np.array_equal(p1, g["c%d_preds_1" % k]) and np.array_equal(p2, g["c%d_preds_2" % k])
        vals, counts = np.unique(p1, return_counts=True)
        kinds.add(("unanimous", cert) if len(vals) == 1 else ("split", pred != vals[counts.argmax()]))
    # every decision branch of PatchCleanser.py:77-94 is pinned
    assert kinds == {("unanimous", True), ("unanimous", False), ("split", True), ("split", False)}


def _checksums(keep):
    H = keep.shape[-1]
    w = torch.arange(H * H, dtype=torch.float64).view(1, 1, H, H) + 1.0
    return keep.sum((1, 2, 3)).numpy(), (keep * w).sum((1, 2, 3)).numpy()


def test_maskwindow_two_patch_tables(golden_patchcleanser):
    """n_patch = 2 (PatchCleanser.py:35-38): mask_set = the 630 pairs, double_mask_set = 36 x 630 triples."""
    g = golden_patchcleanser
    mw = MaskWindow(int(g["H"]), 0.06, 2, device="cpu")
    assert (mw.mask_size, mw.stride, mw.window_size) == tuple(g["np2_params"])
    assert mw.rects.shape == (630, 2, 4) and mw.double_rects.shape == (36 * 630, 3, 4)
    for name, keep in (("np2_single", mw.mask_set), ("np2_double", mw.double_mask_set)):
        cnt, ws = _checksums(keep)
        np.testing.assert_array_equal(cnt, g[name + "_count"])
        np.testing.assert_array_equal(ws, g[name + "_wsum"])
    assert torch.equal(mw.reverse_mask_set, ~mw.mask_set)


def test_maskwindow_single_patch_matches_oracle():
    mw = MaskWindow(56, 0.03, 1, device="cpu")
    assert torch.equal(mw.mask_set, R.single_masks(56, 0.03))
    assert torch.equal(mw.double_mask_set, R.double_masks(56, 0.03))
    assert np.array_equal(mw.rects, masks.single_rects(56, 0.03))


def test_certified_metrics_formulae():
    """main.py:168-184."""
    recs = [PatchCleanserRecord(3, True, np.zeros(36), None), PatchCleanserRecord(1, False, np.zeros(36), None),
            PatchCleanserRecord(2, True, np.zeros(36), None), PatchCleanserRecord(7, True, np.zeros(36), None)]
    y = np.array([3, 1, 5, 6])
    m = certified_metrics(recs, y)
    assert m == {"acc_PC": 50.0, "certified_acc_PC": 25.0, "certified_asr_PC": 50.0}
    m = certified_metrics(recs, y, target=np.array([9, 9, 2, 9]))
    assert m["certified_asr_PC"] == 25.0
