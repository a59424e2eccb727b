"""Mask geometry: product rectangle tables and the oracle's bool masks
against checksums of the reference's MaskWindow (tests/golden/geometry.npz)."""
import numpy as np
import pytest
import torch

from dorpatch_amd import masks
from oracle import restatement as R

RATIOS = (0.015, 0.03, 0.06, 0.12)


def _tag(H, r):
    return "%d_%s" % (H, str(r).replace(".", "p"))


@pytest.mark.parametrize("H", [56, 224, 384])
@pytest.mark.parametrize("r", RATIOS)
def test_window_params(golden_geometry, H, r):
    want = golden_geometry["params_" + _tag(H, r)]
    assert tuple(masks.window_params(H, r)) == tuple(want)
    assert tuple(R.window_geometry(H, r)) == tuple(want)


def test_survey_geometry_table():
    # SURVEY §8 a-10 [ran] values
    assert [masks.window_params(224, r) for r in RATIOS] == [(27, 33, 59), (38, 32, 69), (54, 29, 82), (77, 25, 101)]
    assert [masks.window_params(384, r) for r in RATIOS] == [(47, 57, 103), (66, 54, 119), (94, 49, 142), (133, 42, 174)]


def _checksums(keep):
    H = keep.shape[-1]
    w = torch.arange(H * H, dtype=torch.float64).view(1, 1, H, H) + 1.0
    return keep.sum((1, 2, 3)).numpy(), (keep * w).sum((1, 2, 3)).numpy()


@pytest.mark.parametrize("H", [56, 224])
@pytest.mark.parametrize("r", RATIOS)
def test_mask_sets_match_reference(golden_geometry, H, r):
    tag = _tag(H, r)
    for name, table, oracle_keep in (
            ("single", masks.single_rects(H, r), R.single_masks(H, r)),
            ("double", masks.double_rects(H, r), R.double_masks(H, r))):
        cnt, ws = _checksums(masks.rects_to_bool(table, H))
        np.testing.assert_array_equal(cnt, golden_geometry["%s_count_%s" % (name, tag)])
        np.testing.assert_array_equal(ws, golden_geometry["%s_wsum_%s" % (name, tag)])
        cnt, ws = _checksums(oracle_keep)
        np.testing.assert_array_equal(cnt, golden_geometry["%s_count_%s" % (name, tag)])
        np.testing.assert_array_equal(ws, golden_geometry["%s_wsum_%s" % (name, tag)])


@pytest.mark.parametrize("dropout,n", [(1, 144), (2, 2520)])
def test_universe(dropout, n):
    table = masks.universe_rects(56, dropout)
    assert table.shape == (n, dropout, 4) and table.dtype == np.int32
    assert torch.equal(masks.rects_to_bool(table, 56), R.mask_universe(56, dropout))
    # occluded fraction range quoted in SURVEY §8 a-10 for 224 double masks
    if dropout == 2:
        t224 = masks.rects_to_bool(masks.universe_rects(224, 2)[::37], 224)
        frac = 1.0 - t224.float().mean((1, 2, 3))
        assert 0.05 < frac.min() and frac.max() < 0.42


def test_pad_and_invalid():
    t = masks.pad_rects(masks.single_rects(56, 0.03), 2)
    assert t.shape == (36, 2, 4) and (t[:, 1] == 0).all()
    with pytest.raises(ValueError):
        masks.mask_set_rects(56, 0.03, 0)


@pytest.mark.parametrize("H,dropout", [(56, 1), (56, 2), (224, 2), (384, 2)])
def test_bool_universe_converts_to_an_exactly_equivalent_table(H, dropout):
    """The reference's (n,1,H,W) bool universe (attack.py:83-85) -> rectangle table with the same occluded pixels
    (what DorPatch.collect_failure does when it is handed the reference's tensor)."""
    table = masks.universe_rects(H, dropout)
    if H > 56:
        table = table[np.random.RandomState(0).choice(table.shape[0], 160, replace=False)]
    uni = masks.rects_to_bool(table, H)
    back = masks.bool_to_rects(uni)
    assert back.dtype == np.int32 and back.shape[0] == table.shape[0] and back.shape[1] <= 4
    assert torch.equal(masks.rects_to_bool(back, H), uni)
    assert np.array_equal(masks.bool_to_rects(uni[:, 0]), back)                 # (n, H, W) form


def test_bool_to_rects_edge_cases():
    H = 16
    keep_all = torch.ones(1, 1, H, H, dtype=torch.bool)
    assert not masks.bool_to_rects(keep_all).any()                              # nothing occluded: the empty rectangle
    none = torch.zeros(1, 1, H, H, dtype=torch.bool)
    assert masks.bool_to_rects(none).tolist() == [[[0, H, 0, H]]]
    stair = torch.ones(1, 1, H, H, dtype=torch.bool)
    for i in range(6):                                                          # a staircase is not two windows
        stair[0, 0, i, :i + 1] = False
    with pytest.raises(ValueError):
        masks.bool_to_rects(stair)
    with pytest.raises(ValueError):
        masks.bool_to_rects(torch.ones(1, 1, H, H))                             # not bool
