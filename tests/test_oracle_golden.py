"""Pin the oracle restatement (oracle/restatement.py) against fixtures recorded from the
UNMODIFIED reference (oracle/gen_golden.py): per-step losses, gradients, updated parameters."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import toy_models


def _toy(g):
    return toy_models.NormModel(toy_models.make_toy(gain=float(g["gain"])), toy_models.Normalize())


def _check_steps(g, atol):
    H, S = int(g["H"]), int(g["S"])
    net = _toy(g)
    x = torch.from_numpy(g["x"])
    universe = R.mask_universe(H, int(g["dropout"]) if "dropout" in g else 2)
    lvx = R.local_variance(x)[0].mean(1)
    for n in range(int(g["n_steps"])):
        p = "s%d_" % n
        stage = int(g[p + "stage"])
        keep = universe[torch.from_numpy(g[p + "idx"])]
        dual = dict(keep_dual=universe[torch.from_numpy(g[p + "idx_dual"])]) if (p + "idx_dual") in g else {}
        out = R.eot_step(net, x, torch.from_numpy(g[p + "mask"]), torch.from_numpy(g[p + "pattern"]),
                         torch.tensor([int(g[p + "y"])]), keep, stage=stage,
                         targeted=bool(g["targeted"]) if "targeted" in g else True, n_classes=10, **dual,
                         structured=float(g[p + "structured"]), coeff_group_lasso=float(g[p + "coeff_group_lasso"]),
                         eps=float(g["eps"]), lr=float(g[p + "lr_next"]), local_var_x=lvx)
        np.testing.assert_allclose(out["adv_x"].numpy(), g[p + "adv_x"], atol=atol, rtol=0)
        np.testing.assert_allclose(out["loss_adv"].numpy().reshape(-1), g[p + "loss_adv"], atol=atol, rtol=1e-5)
        np.testing.assert_allclose(out["loss_struc"].item(), g[p + "loss_struc"], rtol=1e-5)
        np.testing.assert_allclose(out["grad_pattern"].numpy(), g[p + "grad_pattern"], atol=atol, rtol=1e-4)
        if stage == 0:
            np.testing.assert_allclose(out["group_lasso"].item(), g[p + "group_lasso"], rtol=1e-5)
            np.testing.assert_allclose(out["density"].item(), g[p + "density"], rtol=1e-4)
            gm, want = out["grad_mask"].numpy(), g[p + "grad_mask"]
            assert np.array_equal(np.isnan(gm), np.isnan(want))
            np.testing.assert_allclose(np.nan_to_num(gm), np.nan_to_num(want), atol=atol, rtol=1e-4)
        # the signed update flips +-lr wherever |grad| ~ ulp: demand near-total agreement
        for key in ("new_pattern", "new_mask"):
            diff = np.abs(out[key].numpy() - g[p + key])
            assert (diff > 1e-6).mean() < 1e-3, (key, n, (diff > 1e-6).mean())


def test_steps_56(golden_steps_56):
    _check_steps(golden_steps_56, atol=1e-6)


def test_steps_224(golden_steps_224):
    _check_steps(golden_steps_224, atol=1e-6)


def test_steps_56_dual(golden_steps_56_dual):
    """attack.py:208-217 (`dual=True`), recorded from the unmodified reference: a second mask set per step."""
    g = golden_steps_56_dual
    assert all(("s%d_idx_dual" % n) in g for n in range(int(g["n_steps"])))
    assert not np.array_equal(g["s0_idx"], g["s0_idx_dual"])
    _check_steps(g, atol=1e-6)


def test_steps_56_dropout1(golden_steps_56_dropout1):
    """`dropout=1` (attack.py:25-31, 83-85), recorded from the unmodified reference: the 144 single-window masks."""
    g = golden_steps_56_dropout1
    assert int(g["dropout"]) == 1 and max(int(g["s%d_idx" % n].max()) for n in range(int(g["n_steps"]))) < 144
    _check_steps(g, atol=1e-6)


def test_steps_56_untargeted(golden_steps_56_untargeted):
    """The untargeted form of CW_loss (attack.py:16-23) and its gradient, recorded from the unmodified reference
    (stage 0 only: the reference raises a TypeError at attack.py:155 entering stage 1 of such a short run)."""
    g = golden_steps_56_untargeted
    assert not bool(g["targeted"]) and int(g["s0_y"]) == int(g["y0"][0]) and "set_target" in str(g["reference_stage1_error"])
    _check_steps(g, atol=1e-6)


def test_final_mask_is_cell_aligned(golden_steps_56):
    m = golden_steps_56["final_mask"]
    assert set(np.unique(m)) <= {0.0, 1.0}
    cells = m.reshape(1, 1, 8, 7, 8, 7)
    assert ((cells.min(axis=(3, 5)) == cells.max(axis=(3, 5)))).all()
    # patch budget 0.12 @56x56 -> floor(3136*0.12/49) = 7 cells
    assert m.sum() <= 7 * 49
