"""GPU parity of every HIP kernel (called through the C ABI) against the CPU oracle.

Tolerances: occlusion apply / argmax / sign bookkeeping are exact; fp32 arithmetic is compared
at rtol 1e-5 (+ small atol) — only summation order differs from the CPU reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dorpatch_amd import _lib, masks, ops  # noqa: E402
from oracle import restatement as R  # noqa: E402

DEV = "cuda:0"


def _rand(*shape, seed=0):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("B,H", [(1, 56), (3, 224), (2, 384)])
@pytest.mark.parametrize("eps", [4.0, 1e4])
def test_blend_matches_clip(B, H, eps):
    x, p, m = _rand(B, 3, H, H, seed=1), _rand(B, 3, H, H, seed=2), _rand(B, 1, H, H, seed=3)
    want = R.clip(m, p, x, eps)
    adv, scale, l2 = ops.blend(m.to(DEV), p.to(DEV), x.to(DEV), eps)
    delta, _, _ = ops.blend(m.to(DEV), p.to(DEV), x.to(DEV), eps, add_x=False)
    l2_want = (m * (p - x)).flatten(1).norm(dim=1)
    np.testing.assert_allclose(l2.cpu().numpy(), l2_want.numpy(), rtol=1e-5)
    np.testing.assert_allclose(scale.cpu().numpy(), (eps / l2_want).clamp(max=1).numpy(), rtol=1e-5)
    np.testing.assert_allclose(delta.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(adv.cpu().numpy(), (want + x).numpy(), rtol=1e-5, atol=1e-7)


def test_blend_zero_delta_gives_unit_scale():
    x = _rand(2, 3, 56, 56)
    m = torch.zeros(2, 1, 56, 56)
    adv, scale, l2 = ops.blend(m.to(DEV), x.clone().to(DEV), x.to(DEV), 4.0)
    assert torch.equal(adv.cpu(), x) and (scale.cpu() == 1).all() and (l2.cpu() == 0).all()


@pytest.mark.parametrize("H,dropout", [(56, 2), (224, 2), (384, 2), (56, 1)])
@pytest.mark.parametrize("normalize", [False, True])
def test_apply_fwd_exact(H, dropout, normalize):
    B, S = 2, 9
    table_np = masks.universe_rects(H, dropout)
    table = ops.upload_table(table_np, DEV)
    rng = np.random.RandomState(H + dropout)
    idx = rng.randint(0, table_np.shape[0], size=(B, S))
    adv = _rand(B, 3, H, H, seed=4)
    keep = R.mask_universe(H, dropout)        # the ORACLE's bool masks (a restatement of PatchCleanser.py:6-59, pinned to the
    assert torch.equal(keep, masks.rects_to_bool(table_np, H))      # reference by geometry.npz), not the product's own
    norm = ops.make_norm([0.5] * 3, [0.5] * 3, 0.5) if normalize else ops.RAW_NORM
    out = ops.apply_fwd(adv.to(DEV), table, torch.from_numpy(idx).int().to(DEV), None, norm).cpu()
    out = out.view(B, S, 3, H, H)
    for b in range(B):
        want = R.occlude(adv[b:b + 1], keep[torch.from_numpy(idx[b])])[0]
        if normalize:
            want = (want - 0.5) / 0.5
        assert torch.equal(out[b], want), (b, (out[b] - want).abs().max())


def test_apply_fwd_dual_and_shared_idx():
    H, B, S = 56, 3, 5
    table_np = masks.universe_rects(H, 2)
    table = ops.upload_table(table_np, DEV)
    keep = masks.rects_to_bool(table_np, H)
    rng = np.random.RandomState(0)
    i1, i2 = rng.randint(0, 2520, S), rng.randint(0, 2520, S)
    adv = _rand(B, 3, H, H, seed=5)
    out = ops.apply_fwd(adv.to(DEV), table, torch.from_numpy(i1).int().to(DEV),
                        torch.from_numpy(i2).int().to(DEV)).cpu().view(B, S, 3, H, H)
    k1, k2 = keep[torch.from_numpy(i1)], keep[torch.from_numpy(i2)]
    want = R.occlude(adv, k1)
    want = want * k2 + 0.5 * ~k2                    # attack.py:218
    assert torch.equal(out, want)


def test_apply_fwd_edge_geometries():
    """Degenerate and maximal inputs of the occlusion kernel: a single sample of a single image, the
    3-window tables of MaskWindow(n_patch=2) (DP_MAX_RECTS - 1), windows clipped at the image border,
    an empty rectangle (nothing occluded), a rectangle covering the whole image (everything = fill), and
    a width with a ragged float4 tail count (W = 20: 5 groups per row)."""
    from dorpatch_amd.patchcleanser import MaskWindow
    H = 56
    adv = _rand(1, 3, H, H, seed=8)
    mw = MaskWindow(H, 0.06, 2)
    assert mw.double_rects.shape == (36 * 630, 3, 4)
    t3 = ops.upload_table(mw.double_rects, DEV)
    pick = np.array([0, 629, 630 * 17 + 311, 36 * 630 - 1])
    keep3 = masks.rects_to_bool(mw.double_rects[pick], H)
    for k, m in enumerate(pick):                                   # S = 1, B = 1
        out = ops.apply_fwd(adv.to(DEV), t3, torch.tensor([int(m)], dtype=torch.int32, device=DEV)).cpu()
        assert torch.equal(out, R.occlude(adv, keep3[k:k + 1])[0])
    special = np.array([[[0, 0, 0, 0]], [[0, H, 0, H]], [[H - 3, H, H - 5, H]]], dtype=np.int32)
    ts = ops.upload_table(special, DEV)
    out = ops.apply_fwd(adv.to(DEV), ts, torch.arange(3, dtype=torch.int32, device=DEV)).cpu()
    assert torch.equal(out[0], adv[0]) and bool((out[1] == 0.5).all())
    want = adv[0].clone()
    want[:, H - 3:, H - 5:] = 0.5
    assert torch.equal(out[2], want)
    Hn, Wn = 12, 20                                                # non-square, W/4 odd
    a2 = _rand(2, 3, Hn, Wn, seed=9)
    t2 = ops.upload_table(np.array([[[2, 7, 3, 18]], [[0, 12, 19, 20]]], dtype=np.int32), DEV)
    o2 = ops.apply_fwd(a2.to(DEV), t2, torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device=DEV)).cpu()
    o2 = o2.view(2, 2, 3, Hn, Wn)
    w0, w1 = a2.clone(), a2.clone()
    w0[:, :, 2:7, 3:18] = 0.5
    w1[:, :, :, 19:20] = 0.5
    assert torch.equal(o2[0, 0], w0[0]) and torch.equal(o2[0, 1], w1[0])
    assert torch.equal(o2[1, 0], w1[1]) and torch.equal(o2[1, 1], w0[1])


def test_invalid_geometry_is_rejected_not_computed():
    """Error convention of the C ABI (hipErrorInvalidValue -> RuntimeError in the host wrapper): a width
    that is not a multiple of 4, more windows than DP_MAX_RECTS, a zero std, B*S mismatch."""
    adv = _rand(1, 3, 8, 6, seed=1).to(DEV)                       # W % 4 != 0
    t = ops.upload_table(np.zeros((1, 1, 4), dtype=np.int32), DEV)
    i0 = torch.zeros(1, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        ops.apply_fwd(adv, t, i0)
    with pytest.raises(AssertionError):
        ops.upload_table(np.zeros((1, 5, 4), dtype=np.int32), DEV)
    ok = _rand(1, 3, 8, 8, seed=2).to(DEV)
    with pytest.raises(RuntimeError):
        ops.apply_fwd(ok, t, i0, None, ops.make_norm([0.5] * 3, [0.5, 0.0, 0.5], 0.5))
    with pytest.raises(AssertionError):
        ops.apply_bwd(torch.zeros(3, 3, 8, 8, device=DEV), t, torch.zeros((2, 2), dtype=torch.int32, device=DEV))
    with pytest.raises(TypeError):
        ops.apply_fwd(ok, t, i0.long())
    with pytest.raises(ValueError):
        ops.apply_fwd(ok.transpose(2, 3), t, i0)


@pytest.mark.parametrize("B,S,H", [(1, 128, 56), (4, 8, 224), (1, 64, 384), (64, 4, 56)])
@pytest.mark.parametrize("normalize", [False, True])
def test_apply_bwd_matches_autograd(B, S, H, normalize):
    table_np = masks.universe_rects(H, 2)
    table = ops.upload_table(table_np, DEV)
    rng = np.random.RandomState(B * S)
    idx = torch.from_numpy(rng.randint(0, 2520, size=(B, S)))
    G = torch.randn(B * S, 3, H, H, generator=torch.Generator().manual_seed(6))
    # oracle: autograd of occlude (+ normalise) w.r.t. adv_x
    keep = masks.rects_to_bool(table_np, H)
    want = torch.empty(B, 3, H, H)
    for b in range(B):
        a = torch.zeros(1, 3, H, H, requires_grad=True)
        o = R.occlude(a, keep[idx[b]])
        if normalize:
            o = (o - 0.5) / 0.5
        o.backward(G[b * S:(b + 1) * S].view(1, S, 3, H, H))
        want[b] = a.grad[0]
    norm = ops.make_norm([0.5] * 3, [0.5] * 3, 0.5) if normalize else ops.RAW_NORM
    got = ops.apply_bwd(G.to(DEV), table, idx.int().to(DEV), None, norm, B=B).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    # accumulate path
    acc = torch.ones(B, 3, H, H, device=DEV)
    ops.apply_bwd(G.to(DEV), table, idx.int().to(DEV), None, norm, B=B, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), want.numpy() + 1, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C", [10, 1000])
def test_cw_loss_grad_pred(C):
    B, S = 3, 7
    N = B * S
    logits = torch.randn(N, C, generator=torch.Generator().manual_seed(7)) * 2
    logits[0, 3] = 9.0          # satisfied margin for a targeted row with y = 3
    y = torch.tensor([3, 1, 5])
    flags = [True, False, True]
    conf, up = 0.1, 1.0 / S
    lg = logits.clone().requires_grad_(True)
    rows = [R.cw_loss(lg[b * S:(b + 1) * S], y[b].repeat(S), C, flags[b], conf) for b in range(B)]
    want = torch.cat(rows)
    (want.sum() * up).backward()
    loss, dl, pred = ops.cw_loss(logits.to(DEV), y.to(DEV), torch.tensor(flags).int().to(DEV), S, conf, up)
    np.testing.assert_allclose(loss.cpu().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dl.cpu().numpy(), lg.grad.numpy(), rtol=1e-6, atol=0)
    assert torch.equal(pred.cpu().long(), logits.argmax(-1))
    assert torch.equal(ops.argmax(logits.to(DEV)).cpu().long(), logits.argmax(-1))
    assert (want == 0).any() and (want > 0).any()


@pytest.mark.parametrize("B,H", [(2, 56), (1, 224), (1, 384), (2, 40)])
def test_local_variance_and_struct_loss(B, H):
    x, a = _rand(B, 3, H, H, seed=8), _rand(B, 3, H, H, seed=9)
    lv_want = R.local_variance(x)[0].mean(1)
    lv = ops.local_variance(x.to(DEV))
    np.testing.assert_allclose(lv.cpu().numpy(), lv_want.numpy(), rtol=1e-6, atol=1e-7)
    want = R.struct_loss(a, lv_want)
    got = ops.struct_loss(a.to(DEV), lv)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-5)


@pytest.mark.parametrize("B,H", [(2, 56), (1, 224), (1, 384)])
def test_mask_stats(B, H):
    m = _rand(B, 1, H, H, seed=10)
    m[0, 0, 7:14, 14:21] = 0.0          # an all-zero group-lasso cell
    cell, wsum, gl, dens = ops.mask_stats(m.to(DEV), 7, H // 8)
    import torch.nn.functional as F
    cell_want = F.conv2d(m ** 2, torch.ones(1, 1, 7, 7), stride=7)[:, 0]
    wsum_want = F.conv2d(m, torch.ones(1, 1, H // 8, H // 8), stride=H // 8)[:, 0]
    np.testing.assert_allclose(cell.cpu().numpy(), cell_want.numpy(), rtol=1e-5)
    assert cell.cpu()[0, 1, 2] == 0
    np.testing.assert_allclose(wsum.cpu().numpy(), wsum_want.numpy(), rtol=1e-5)
    np.testing.assert_allclose(gl.cpu().numpy(), R.group_lasso(m).numpy(), rtol=1e-5)
    np.testing.assert_allclose(dens.cpu().numpy(), R.density_loss(m).numpy(), rtol=1e-4)


def _grads_case(B, H, stage, seed):
    """Full gradient chain of one step: oracle autograd vs dp_project_update."""
    from oracle import toy_models
    net = toy_models.NormModel(toy_models.make_toy(gain=1.0), toy_models.Normalize())
    S = 6
    x, p, m = _rand(B, 3, H, H, seed=seed), _rand(B, 3, H, H, seed=seed + 1), _rand(B, 1, H, H, seed=seed + 2)
    if stage == 0:
        m[0, 0, 0:7, 0:7] = 0.0         # frozen cell -> NaN gradient
    else:
        m = (m > 0.8).float()
    y = torch.arange(B) % 10
    universe = R.mask_universe(H, 2)
    idx = torch.from_numpy(np.random.RandomState(seed).randint(0, 2520, size=(B, S)))
    keep = universe[idx]                # (B,S,1,H,H)
    structured = [1e-3 * (b + 1) for b in range(B)]
    coeff = [1e-5 * (b + 2) for b in range(B)]
    lr = [0.01 * (b + 1) for b in range(B)]
    o = R.eot_step(net, x, m, p, y, keep, stage=stage, targeted=True, n_classes=10, structured=structured,
                   coeff_group_lasso=coeff, eps=4.0, lr=lr)
    # d loss_adv.mean / d adv_x from the oracle -> g_adv input of the fused kernel
    adv = o["adv_x"].clone().requires_grad_(True)
    masked = (adv[:, None] * keep + 0.5 * ~keep).reshape(-1, 3, H, H)
    lg = net(masked)
    la = torch.stack([R.cw_loss(lg[b * S:(b + 1) * S], y[b].repeat(S), 10, True, 0.1) for b in range(B)])
    la.mean(1).sum().backward()
    return dict(x=x, p=p, m=m, y=y, idx=idx, o=o, g_adv=adv.grad, structured=structured, coeff=coeff, lr=lr)


@pytest.mark.parametrize("stage", [0, 1])
@pytest.mark.parametrize("B,H", [(2, 56), (1, 224)])
def test_project_update_grads_and_update(stage, B, H):
    c = _grads_case(B, H, stage, seed=20 + stage)
    d = lambda t: t.to(DEV).contiguous()
    x, p, m = d(c["x"]), d(c["p"]), d(c["m"])
    adv, scale, _ = ops.blend(m, p, x, 4.0)
    lv = ops.local_variance(x)
    cell, wsum, gl, dens = ops.mask_stats(m, 7, H // 8)
    f32 = lambda v: torch.tensor(v, dtype=torch.float32, device=DEV)
    kw = dict(stage=stage, coeff_gl=f32(c["coeff"]), cell_sumsq=cell, win_sum=wsum, unit=7, win=H // 8,
              density=1e-3)
    gp, gm = ops.project_update(x, adv, lv, d(c["g_adv"]), scale, f32(c["structured"]), p, m,
                                do_update=False, want_grads=True, **kw)
    o = c["o"]
    gp_want, gm_want = o["grad_pattern"].numpy(), o["grad_mask"].numpy()
    scale_g = np.abs(gp_want).max()
    np.testing.assert_allclose(gp.cpu().numpy(), gp_want, rtol=1e-4, atol=1e-6 * scale_g)
    assert np.array_equal(np.isnan(gm.cpu().numpy()), np.isnan(gm_want))
    if stage == 0:
        assert np.isnan(gm_want).sum() == 49
        # the density term is (window_sum - mean) * 3e-5 with window sums ~ H*H/128: the reference's own
        # fp32 accumulation of a 28x28 window is only good to ~3e-4 absolute, i.e. ~1e-8 in the gradient
        np.testing.assert_allclose(np.nan_to_num(gm.cpu().numpy()), np.nan_to_num(gm_want), rtol=1e-4,
                                   atol=1e-4 * np.nanmax(np.abs(gm_want)))
    assert torch.equal(p.cpu(), c["p"]) and torch.equal(m.cpu(), c["m"])       # do_update=False
    # now the signed update + best-copy
    best_p, best_m = torch.zeros_like(p), torch.zeros_like(m)
    save = torch.tensor([1] + [0] * (B - 1), dtype=torch.int32, device=DEV)
    p0, m0 = p.clone(), m.clone()
    ops.project_update(x, adv, lv, d(c["g_adv"]), scale, f32(c["structured"]), p, m, lr=f32(c["lr"]),
                       save_best=save, best_pattern=best_p, best_mask=best_m, do_update=True, **kw)
    assert torch.equal(best_p[0], p0[0]) and (best_p[1:] == 0).all()
    if stage == 0:
        assert torch.equal(best_m[0], m0[0])
    else:
        assert torch.equal(m, m0) and (best_m == 0).all()          # mask frozen in stage 1
    for got, want in ((p.cpu().numpy(), o["new_pattern"].numpy()), (m.cpu().numpy(), o["new_mask"].numpy())):
        flips = np.abs(got - want) > 1e-6
        assert flips.mean() < 2e-3, flips.mean()     # sign(ulp-noise) pixels only
        assert np.abs(got - want).max() <= 2 * max(c["lr"]) + 1e-6
    assert p.min() >= 0 and p.max() <= 1


@pytest.mark.parametrize("stage", [0, 1])
@pytest.mark.parametrize("B,H,W", [(2, 56, 56), (1, 224, 224), (1, 384, 384), (3, 40, 36), (2, 33, 100), (1, 7, 8)])
def test_project_update_wide_lanes_equal_the_scalar_kernel(stage, B, H, W):
    """dp_project_update's 16-byte-lane kernel (one lane = 4 pixels x 3 channels, 32 x 32 tiles) against the 4-byte-lane
    kernel (DP_DEBUG_UPDATE_VARIANT = 1) it replaces: gradients incl. the NaN cells, best-so-far copies and updated
    parameters bit for bit, on tile-aligned sizes (224 = 7 tiles), ragged ones (36, 100, 40 rows) and a single cell."""
    from dorpatch_amd._lib import DP_DEBUG_UPDATE_VARIANT as KNOB
    g = torch.Generator().manual_seed(100 * stage + H + W)
    r = lambda *shape: torch.rand(*shape, generator=g).to(DEV)
    x, adv, p, m = r(B, 3, H, W), r(B, 3, H, W), r(B, 3, H, W), r(B, 1, H, W)
    adv[0, :, : H // 2] = adv[0, :, : H // 2].round()          # ties a == b and zero differences: the sign / min branches
    if stage == 0:
        m[0, 0, 0:7, 0:7] = 0.0                                 # frozen cell -> NaN
    g_adv = (torch.randn(B, 3, H, W, generator=g) * 1e-3).to(DEV)
    g_adv[:, :, ::3, ::5] = 0.0                                 # sign(0) = 0 pixels
    lv = r(B, H, W) * 0.5
    f32 = lambda v: torch.tensor(v, dtype=torch.float32, device=DEV)
    unit, win = 7, max(1, W // 8)
    ncy, ncx, nwy, nwx = (H - unit) // unit + 1, (W - unit) // unit + 1, (H - win) // win + 1, (W - win) // win + 1
    cell, wsum = r(B, ncy, ncx) + 0.1, r(B, nwy, nwx) * win * win
    if stage == 0:
        cell[0, 0, 0] = 0.0
    kw = dict(stage=stage, coeff_gl=f32([1e-5 * (b + 2) for b in range(B)]), cell_sumsq=cell.contiguous(),
              win_sum=wsum.contiguous(), unit=unit, win=win, density=1e-3)
    scale, structured = f32([0.5 + 0.1 * b for b in range(B)]), f32([1e-3 * (b + 1) for b in range(B)])
    lr = f32([0.01 * (b + 1) for b in range(B)])
    save = torch.tensor([1] + [0] * (B - 1), dtype=torch.int32, device=DEV)
    outs = []
    try:
        for variant in (1, 0):
            ops.debug_set(KNOB, variant)
            pp, mm = p.clone(), m.clone()
            bp, bm = torch.full_like(p, -1.0), torch.full_like(m, -1.0)
            gp, gm = ops.project_update(x, adv, lv, g_adv, scale, structured, pp, mm, do_update=False, want_grads=True, **kw)
            assert torch.equal(pp, p) and torch.equal(mm, m)
            ops.project_update(x, adv, lv, g_adv, scale, structured, pp, mm, lr=lr, save_best=save, best_pattern=bp,
                               best_mask=bm, do_update=True, **kw)
            outs.append([t.cpu() for t in (gp, gm, pp, mm, bp, bm)])
    finally:
        ops.debug_set(KNOB, 0)
    names = ("grad_pattern", "grad_mask", "pattern", "mask", "best_pattern", "best_mask")
    for name, a, b in zip(names, outs[0], outs[1]):
        assert torch.equal(torch.isnan(a), torch.isnan(b)), name
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), (name, (a - b).abs().max())
    if stage == 0 and H >= 7 and W >= 8:
        assert torch.isnan(outs[0][1]).any()
    assert not torch.equal(outs[0][2], p.cpu())                 # the update did move the pattern


@pytest.mark.parametrize("B,S,H", [(3, 5, 56), (2, 4, 224), (9, 3, 40)])
def test_apply_fwd_launch_orders_are_equivalent(B, S, H):
    """dp_apply_fwd's A/B launch order (DP_DEBUG_APPLY_ORDER = 1: workgroups walk XCD by XCD, a unit's samples adjacent
    on one XCD, the source tile fetched from HBM once — less traffic, measured slower, not the default) against the
    3-D grid.  Same bytes either way, also when the number of (image, tile) units is not a multiple of 8 (the padded
    workgroups of the 1-D launch exit)."""
    from dorpatch_amd._lib import DP_DEBUG_APPLY_ORDER as KNOB
    x = _rand(B, 3, H, H, seed=3).to(DEV)
    table = ops.upload_table(masks.universe_rects(H, 2), DEV)
    idx = torch.from_numpy(np.random.RandomState(1).randint(0, 2520, size=(B, S))).int().to(DEV)
    idx2 = torch.from_numpy(np.random.RandomState(2).randint(0, 2520, size=(B, S))).int().to(DEV)
    norm = ops.make_norm([0.5] * 3, [0.5] * 3, 0.5)
    outs = []
    try:
        for order in (1, 0):
            ops.debug_set(KNOB, order)
            outs.append((ops.apply_fwd(x, table, idx, None, norm).cpu(), ops.apply_fwd(x, table, idx, idx2, ops.RAW_NORM).cpu()))
    finally:
        ops.debug_set(KNOB, 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_debug_knobs_reject_unknown_values():
    with pytest.raises(RuntimeError):
        ops.debug_set(99, 0)
    with pytest.raises(RuntimeError):
        ops.debug_set(2, 5)


CONV3X3_CASES = [   # (N, C, O, S, emulation-sized)
    (2, 16, 64, 28, True),      # 3.5 tiles: tiles that start mid-plane, cross into the next image, a ragged last tile;
                                # 2 K-chunks: the double buffer is exercised
    (10, 8, 128, 7, True),      # 1.1 tiles of 9.1 planes each (scalar staging), two output-channel groups, ONE K-chunk
    (1, 16, 64, 56, False),     # one plane = 7 tiles (float4 staging, pitch 64)
    (5, 16, 64, 14, False),     # 2.2 tiles of 2.3 planes each (float2 staging)
    (3, 64, 64, 56, False), (5, 128, 128, 28, False), (9, 256, 256, 14, False), (37, 512, 512, 7, False),   # ResNetV2-50's four
]


@pytest.mark.parametrize("N,C,O,S,small", CONV3X3_CASES)
def test_conv3x3_on_the_matrix_cores_matches_conv2d(N, C, O, S, small):
    """dp_conv3x3_fwd (direct implicit GEMM on v_mfma_f32_32x32x2_f32: the backbone's 3x3 / 1 convolutions) against
    F.conv2d, forward and — the same entry point on transposed + flipped weights — input gradient: exact-f32 arithmetic,
    another summation order -> 1e-5 of the output scale.  Structured inputs catch layout slips a random tensor would blur:
    one-hot weights (each output channel copies ONE shifted input channel: exact equality, including the zero padding at
    all four borders of every image and at the seams between the images a tile spans), an asymmetric ramp image.  Two
    small cases are what the CPU emulation affords by default (DORPATCH_EMU_FULL=1: every case but the four full-width
    ones); everything runs on the GPU."""
    import os
    if DEV == "cpu" and not small and not (os.environ.get("DORPATCH_EMU_FULL", "0") == "1" and C <= 16):
        pytest.skip("through the fibre emulation this case takes minutes: GPU (or DORPATCH_EMU_FULL=1 for the narrow ones)")
    g = torch.Generator().manual_seed(C + S)
    x = torch.randn(N, C, S, S, generator=g)
    x[0] += torch.arange(float(S)).view(1, S, 1) * 0.1 + torch.arange(float(S)).view(1, 1, S) * 0.01
    w = torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    want = F.conv2d(x, w, padding=1)
    got = ops.conv3x3_fwd(x.to(DEV).contiguous(), ops.pack_conv3x3_weights(w).to(DEV)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    # one-hot weights: output channel o = input channel (5 o + 3) % C shifted by tap (o % 3, (o // 3) % 3) — exact
    w1 = torch.zeros(O, C, 3, 3)
    for o in range(O):
        w1[o, (5 * o + 3) % C, o % 3, (o // 3) % 3] = 1.0
    got1 = ops.conv3x3_fwd(x.to(DEV).contiguous(), ops.pack_conv3x3_weights(w1).to(DEV)).cpu()
    assert torch.equal(got1, F.conv2d(x, w1, padding=1))
    if C % 64 == 0:        # input gradient = the same kernel on dy with transposed, flipped weights
        dy = torch.randn(N, O, S, S, generator=g)
        want_dx = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                      (True, False, False))[0]
        got_dx = ops.conv3x3_fwd(dy.to(DEV).contiguous(), ops.pack_conv3x3_weights(w, transpose=True).to(DEV)).cpu()
        np.testing.assert_allclose(got_dx.numpy(), want_dx.numpy(), rtol=0, atol=1e-5 * float(want_dx.abs().max()))


CONV3X3_FOLD_CASES = [   # (N, C, O, S, emulation-sized)
    (2, 32, 64, 28, True), (3, 32, 64, 14, True), (1, 32, 64, 56, False),
    (3, 64, 64, 56, False), (5, 128, 128, 28, False), (9, 256, 256, 14, False),
]


@pytest.mark.parametrize("N,C,O,S,small", CONV3X3_FOLD_CASES)
def test_conv3x3_with_folded_groupnorm_is_bit_identical_to_the_two_kernels(N, C, O, S, small):
    """dp_gn_stats + dp_conv3x3_gn_fwd == dp_gn_relu_fwd + dp_conv3x3_fwd, bit for bit: the fold applies the
    GroupNorm kernel's own expression while staging, and the zero padding (image borders, the rows between the images a tile
    spans) pads the NORMALISED activation — a beta > 0 would leak into the halo if the transform touched it."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    G = 32
    g = torch.Generator().manual_seed(C + S)
    x = (torch.randn(N, C, S, S, generator=g) * 1.5 + 0.3).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.2 + 1.0).to(DEV)        # mostly positive: relu(beta) != 0 in a polluted halo
    wt = ops.pack_conv3x3_weights(torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(DEV)
    y, mean, rstd, _ = ops.gn_relu_fwd(x, gamma, beta, G, 1e-5)
    want = ops.conv3x3_fwd(y, wt)
    mean2, rstd2, ab, _ = ops.gn_stats(x, gamma, beta, G, 1e-5)
    assert torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    got = ops.conv3x3_fwd(x, wt, ab=ab)
    assert torch.equal(got, want)


CONV3X3_FLAT_CASES = [   # (N, C, O, S, fold groups or 0, emulation-sized)
    (3, 16, 64, 14, 16, True),      # 1.3 tiles of 2.3 planes each: halo items before the batch, seams inside the tile, ragged end
    (11, 8, 64, 7, 0, True),        # flat mode: 9 whole images + a ragged second tile, ONE chunk
    (1, 16, 64, 28, 0, False),      # 1.75 tiles inside one plane: the halo is the same image's rows
    (3, 8, 64, 12, 0, True),        # a 384-input side: 432 pixels, one ragged tile
    (1, 8, 64, 24, 0, False), (1, 8, 64, 48, 0, False), (1, 32, 64, 96, 32, False), (2, 8, 64, 56, 0, False),
    (9, 256, 256, 14, 32, False), (37, 512, 512, 7, 0, False), (5, 128, 128, 28, 32, False), (2, 128, 128, 48, 32, False),
]


@pytest.mark.parametrize("N,C,O,S,G,small", CONV3X3_FLAT_CASES)
def test_conv3x3_flat_kernel_matches_conv2d_and_the_row_kernel(N, C, O, S, G, small):
    """k_conv3x3_flat (round 5: flat LDS image, masked taps — every side, incl. the 384-input ones) behind dp_conv3x3_fwd
    with DP_DEBUG_CONV3X3_VARIANT = 2: against F.conv2d at 1e-5 of the output scale; one-hot weights exact (every tap's
    mask at all four borders, the seams between images, the halo before the first / after the last pixel of the batch);
    where k_conv3x3_mfma takes the side too: bit-identical to it (same k-walk); GroupNorm fold bit-identical to normalising
    first (a positive beta would leak through a tap that is not masked)."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    g = torch.Generator().manual_seed(C + S)
    x = torch.randn(N, C, S, S, generator=g)
    x[0] += torch.arange(float(S)).view(1, S, 1) * 0.1 + torch.arange(float(S)).view(1, 1, S) * 0.01
    w = torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    want = F.conv2d(x, w, padding=1)
    xd, wt = x.to(DEV).contiguous(), ops.pack_conv3x3_weights(w).to(DEV)
    w1 = torch.zeros(O, C, 3, 3)
    for o in range(O):
        w1[o, (5 * o + 3) % C, o % 3, (o // 3) % 3] = 1.0
    try:
        ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 2)
        got = ops.conv3x3_fwd(xd, wt)
        got1 = ops.conv3x3_fwd(xd, ops.pack_conv3x3_weights(w1).to(DEV)).cpu()
        if G:
            xr = (x * 1.5 + 0.3).to(DEV).contiguous()
            gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
            beta = (torch.randn(C, generator=g) * 0.2 + 1.0).to(DEV)
            y, mean, rstd, _ = ops.gn_relu_fwd(xr, gamma, beta, G, 1e-5)
            _, _, ab, _ = ops.gn_stats(xr, gamma, beta, G, 1e-5)
            assert torch.equal(ops.conv3x3_fwd(xr, wt, ab=ab), ops.conv3x3_fwd(y, wt))
    finally:
        ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 0)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    assert torch.equal(got1, F.conv2d(x, w1, padding=1))
    if S in (56, 28, 14, 7):
        try:
            ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 1)
            rows = ops.conv3x3_fwd(xd, wt)
        finally:
            ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 0)
        assert torch.equal(got, rows)


CONV3X3_TILE_CASES = [   # (N, C, O, S, fold groups or 0, emulation-sized)
    (3, 16, 64, 14, 16, True),      # 588 pixels: 1.3 / 4.6 / 9.2 tiles, seams of images inside every tile size, ragged ends
    (5, 8, 64, 7, 0, True),         # flat mode: 9 / 2 / 1 whole images per tile
    (2, 16, 128, 28, 16, False), (3, 8, 64, 12, 0, False), (1, 16, 64, 24, 16, False),
    (9, 256, 256, 14, 32, False), (37, 512, 512, 7, 0, False), (5, 128, 128, 28, 32, False),
]


@pytest.mark.parametrize("N,C,O,S,G,small", CONV3X3_TILE_CASES)
def test_conv3x3_flat_pixel_tiles_are_bit_identical(N, C, O, S, G, small):
    """Round 6 (VERDICT r5 items 1 + 6): k_conv3x3_flat with 128- and 64-pixel tiles (K walked in 4-channel chunks) for
    the batches whose planes give too few 448-pixel tiles — the same k-walk per output element, so plain and folded
    launches must give the bits of the 448-pixel tile (which the test above pins to F.conv2d and to k_conv3x3_mfma), and
    the launcher's own pick as well.  One-hot weights once more per tile: exact against F.conv2d."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    g = torch.Generator().manual_seed(C + S + N)
    x = torch.randn(N, C, S, S, generator=g)
    w = torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    xd, wt = x.to(DEV).contiguous(), ops.pack_conv3x3_weights(w).to(DEV)
    w1 = torch.zeros(O, C, 3, 3)
    for o in range(O):
        w1[o, (5 * o + 3) % C, o % 3, (o // 3) % 3] = 1.0
    wt1, want1 = ops.pack_conv3x3_weights(w1).to(DEV), F.conv2d(x, w1, padding=1)
    if G:
        gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
        beta = (torch.randn(C, generator=g) * 0.2 + 1.0).to(DEV)
        xr = (x * 1.5 + 0.3).to(DEV).contiguous()
        _, _, ab, _ = ops.gn_stats(xr, gamma, beta, G, 1e-5)
    outs = {}
    try:
        for variant in (2, 1 << 4, 3 << 4, 4 << 4, 0):       # flat 448 (reference), forced 448 / 128 / 64, the launcher's rule
            ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, variant)
            got = [ops.conv3x3_fwd(xd, wt).cpu()]
            if G:
                got.append(ops.conv3x3_fwd(xr, wt, ab=ab).cpu())
            outs[variant] = got
            assert torch.equal(ops.conv3x3_fwd(xd, wt1).cpu(), want1), variant
    finally:
        ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 0)
    for variant, got in outs.items():
        for a, b in zip(got, outs[2]):
            assert torch.equal(a, b), "variant %d differs from the 448-pixel flat tile" % variant


CONV3X3_WINO_CASES = [   # (N, C, O, S, emulation-sized)
    (2, 8, 64, 14, True),       # 98 tiles: 3 full blocks of 32 + a ragged one, blocks that cross the image boundary, 2 K-chunks
    (3, 8, 64, 7, True),        # odd side: 4 x 4 tiles cover 8 x 8, the overhang is not stored; TWO chunks (the minimum)
    (1, 8, 128, 28, False),     # two output-channel groups
    (3, 64, 64, 56, False), (5, 128, 128, 28, False), (9, 256, 256, 14, False), (20, 512, 512, 7, False),   # ResNetV2-50's four
    (1, 64, 64, 96, False), (2, 128, 128, 48, False), (3, 8, 64, 24, False), (5, 8, 64, 12, True),          # 384 x 384 inputs
]


@pytest.mark.parametrize("N,C,O,S,small", CONV3X3_WINO_CASES)
def test_conv3x3_winograd_on_the_matrix_cores_matches_conv2d(N, C, O, S, small):
    """dp_conv3x3_wino_fwd (round 6: Winograd F(2x2, 3x3), the 16 position GEMMs on v_mfma_f32_32x32x2_f32) against
    F.conv2d and — on transposed + flipped weights — ATen's input gradient.  Winograd sums 16 transformed products where
    the direct form sums 9 plain ones: fp32 round-off of another shape, so the bound is 2e-5 of the output scale (measured
    ~1e-6; the direct kernels are held to 1e-5) and the one-hot check (every output channel copies ONE shifted input
    channel: padding at all four borders of every image, the seams between the images a tile block spans, the odd side's
    overhang) is to 2e-6 of the input scale instead of exact.  With the GroupNorm fold: bit-identical to normalising first
    (same kernel, same patch values; the padding pads the NORMALISED activation).  Deterministic: two launches, same bits."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    g = torch.Generator().manual_seed(C + S)
    x = torch.randn(N, C, S, S, generator=g)
    x[0] += torch.arange(float(S)).view(1, S, 1) * 0.1 + torch.arange(float(S)).view(1, 1, S) * 0.01
    w = torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    want = F.conv2d(x, w, padding=1)
    xd, wt = x.to(DEV).contiguous(), ops.pack_conv3x3_wino_weights(w).to(DEV)
    got = ops.conv3x3_wino_fwd(xd, wt)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))
    assert torch.equal(got, ops.conv3x3_wino_fwd(xd, wt))
    w1 = torch.zeros(O, C, 3, 3)
    for o in range(O):
        w1[o, (5 * o + 3) % C, o % 3, (o // 3) % 3] = 1.0
    got1 = ops.conv3x3_wino_fwd(xd, ops.pack_conv3x3_wino_weights(w1).to(DEV)).cpu()
    np.testing.assert_allclose(got1.numpy(), F.conv2d(x, w1, padding=1).numpy(), rtol=0, atol=2e-6 * float(x.abs().max()))
    if C % 64 == 0:        # input gradient = the same kernel on dy with transposed, flipped weights
        dy = torch.randn(N, O, S, S, generator=g)
        want_dx = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                      (True, False, False))[0]
        got_dx = ops.conv3x3_wino_fwd(dy.to(DEV).contiguous(), ops.pack_conv3x3_wino_weights(w, transpose=True).to(DEV)).cpu()
        np.testing.assert_allclose(got_dx.numpy(), want_dx.numpy(), rtol=0, atol=2e-5 * float(want_dx.abs().max()))
    G = 32 if C % 32 == 0 else C // 2        # GroupNorm fold (any side: the staging loads are scalar)
    if (C // G) * S * S % 4 == 0:
        xr = (x * 1.5 + 0.3).to(DEV).contiguous()
        gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
        beta = (torch.randn(C, generator=g) * 0.2 + 1.0).to(DEV)        # mostly positive: relu(beta) != 0 in a polluted halo
        y, mean, rstd, _ = ops.gn_relu_fwd(xr, gamma, beta, G, 1e-5)
        _, _, ab, _ = ops.gn_stats(xr, gamma, beta, G, 1e-5)
        assert torch.equal(ops.conv3x3_wino_fwd(xr, wt, ab=ab), ops.conv3x3_wino_fwd(y, wt))


@pytest.mark.parametrize("C,S", [(64, 56), (128, 28), (256, 14), (512, 7)])
def test_conv3x3_winograd_at_the_bench_size_agrees_with_the_direct_kernel(C, S):
    """BASELINE configs[1]'s micro-batch (N = 512) on ResNetV2-50's four stride-1 3x3 shapes — sizes no CPU reference finishes
    in seconds, so through size-independent properties: the Winograd kernel against the direct MFMA kernel (another
    algorithm, other index arithmetic, the same weights) to 2e-5 of the output scale everywhere; the LAST image against
    F.conv2d on the CPU (the end of the grid: ragged last block, largest offsets); linearity in the input; same bits twice."""
    if DEV == "cpu":
        pytest.skip("bench-sized: GPU only")
    N = 512
    g = torch.Generator().manual_seed(S)
    x = torch.randn(N, C, S, S, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    xd = x.to(DEV)
    ww, wd = ops.pack_conv3x3_wino_weights(w).to(DEV), ops.pack_conv3x3_weights(w).to(DEV)
    got = ops.conv3x3_wino_fwd(xd, ww)
    ref = ops.conv3x3_fwd(xd, wd)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale
    np.testing.assert_allclose(got[-1:].cpu().numpy(), F.conv2d(x[-1:], w, padding=1).numpy(), rtol=0, atol=2e-5 * scale)
    assert torch.equal(got, ops.conv3x3_wino_fwd(xd, ww))
    x2 = torch.randn(N, C, S, S, generator=g).to(DEV)
    lin = ops.conv3x3_wino_fwd((0.5 * xd + x2).contiguous(), ww)
    assert float((lin - (0.5 * got + ops.conv3x3_wino_fwd(x2, ww))).abs().max()) <= 2e-5 * scale


CONV3X3S2_BWD_CASES = [   # (N, O = channels of dy, C = channels of dx, side of dy, emulation-sized)
    (2, 16, 64, 14, True),      # 392 pixels: one ragged tile spanning both images; ONE chunk in class (0,0), 4 in class (1,1)
    (10, 16, 64, 7, True),      # flat mode: 9 whole images + a ragged second tile; 1 - 4 chunks per class
    (1, 16, 128, 28, False),    # 1.75 tiles inside one plane, two channel groups
    (3, 16, 64, 12, False),     # a 384-input side
    (3, 128, 128, 28, False), (5, 256, 256, 14, False), (20, 512, 512, 7, False), (2, 128, 128, 48, False),
]


@pytest.mark.parametrize("N,O,C,S,small", CONV3X3S2_BWD_CASES)
def test_conv3x3_stride2_input_gradient_on_the_matrix_cores(N, O, C, S, small):
    """dp_conv3x3s2_bwd (round 5: the input gradient of the three stride-2 3x3 convolutions as masked parity-class walks, both
    forms: two column classes per workgroup with interleaved 8-byte stores / one class per workgroup) against ATen's
    convolution_backward at 1e-5 of the gradient scale; one-hot weights exact (each dx channel
    receives ONE dy channel through ONE tap: the scatter to (2a + pr, 2b + pc), the bottom / right masks and the rows /
    columns no tap reaches are all visible); every element of dx is written (the output starts as NaN)."""
    import os
    if DEV == "cpu" and not small and not (os.environ.get("DORPATCH_EMU_FULL", "0") == "1" and O <= 16):
        pytest.skip("through the fibre emulation this case takes minutes: GPU (or DORPATCH_EMU_FULL=1 for the narrow ones)")
    g = torch.Generator().manual_seed(C + S)
    dy = torch.randn(N, O, S, S, generator=g)
    dy[0] += torch.arange(float(S)).view(1, S, 1) * 0.1 + torch.arange(float(S)).view(1, 1, S) * 0.01
    w = torch.randn(O, C, 3, 3, generator=g) / (3.0 * O ** 0.5)
    x_ref = torch.zeros(N, C, 2 * S, 2 * S)

    def ref(wq):
        return torch.ops.aten.convolution_backward(dy, x_ref, wq, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                                   (True, False, False))[0]
    dyd = dy.to(DEV).contiguous()
    got = ops.conv3x3s2_bwd(dyd, ops.pack_conv3x3s2_dgrad_weights(w).to(DEV), C).cpu()
    want = ref(w)
    assert tuple(got.shape) == (N, C, 2 * S, 2 * S) and not torch.isnan(got).any()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    w1 = torch.zeros(O, C, 3, 3)
    for c in range(C):
        w1[(5 * c + 3) % O, c, c % 3, (c // 3) % 3] = 1.0
    got1 = ops.conv3x3s2_bwd(dyd, ops.pack_conv3x3s2_dgrad_weights(w1).to(DEV), C).cpu()
    assert torch.equal(got1, ref(w1))
    # the one-class-per-workgroup form: each class's own k-walk in the same order -> the same bits
    got4 = ops.conv3x3s2_bwd(dyd, ops.pack_conv3x3s2_dgrad_weights(w, pairs=False).to(DEV), C, pairs=False).cpu()
    assert not torch.isnan(got4).any() and torch.equal(got4, got)


STEM_CONV_CASES = [(1, 16, True), (2, 6, True), (3, 224, False), (2, 10, False)]      # (N, H, emulation-sized); W = 224


@pytest.mark.parametrize("N,H,small", STEM_CONV_CASES)
def test_stem_convolution_on_the_matrix_cores_matches_conv2d(N, H, small):
    """dp_stem_conv_fwd (round 5: 3 -> 64, 7x7 / stride 2 / pad 3 with the whole K in LDS, column-parity de-interleaved
    rows) against F.conv2d at 1e-5 of the output scale; one-hot weights exact: output channel o copies input channel
    o % 3 through tap (o % 7, (o // 7) % 7) — every tap, the padding at all four borders, the ragged last tile (H / 2 not a
    multiple of 4 rows)."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, 3, H, 224, generator=g)
    x[0] += torch.arange(float(H)).view(1, H, 1) * 0.1 + torch.arange(224.0).view(1, 1, 224) * 0.01
    w = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    want = F.conv2d(x, w, stride=2, padding=3)
    xd = x.to(DEV).contiguous()
    got = ops.stem_conv_fwd(xd, ops.pack_stem_weights(w).to(DEV)).cpu()
    assert tuple(got.shape) == (N, 64, H // 2, 112)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    w1 = torch.zeros(64, 3, 7, 7)
    for o in range(64):
        w1[o, o % 3, o % 7, (o // 7) % 7] = 1.0
    got1 = ops.stem_conv_fwd(xd, ops.pack_stem_weights(w1).to(DEV)).cpu()
    assert torch.equal(got1, F.conv2d(x, w1, stride=2, padding=3))


CONV3X3S2_CASES = [   # (N, C, O, INPUT side, emulation-sized)
    (2, 16, 64, 28, True),      # 392 output pixels: one ragged tile spanning both images, float4 staging, 8 K-chunks
    (10, 8, 128, 14, True),     # 490 output pixels = 1.1 tiles of 9.1 planes each (float2 staging), two output-channel groups
    (1, 8, 64, 56, False),      # one plane = 1.75 tiles (a tile that starts mid-plane)
    (3, 128, 128, 56, False), (5, 256, 256, 28, False), (20, 512, 512, 14, False),      # ResNetV2-50's three
]


@pytest.mark.parametrize("tile", [0, 1, 3, 4], ids=["auto", "448px", "128px", "64px"])
@pytest.mark.parametrize("N,C,O,S,small", CONV3X3S2_CASES)
def test_conv3x3_stride2_on_the_matrix_cores_matches_conv2d(N, C, O, S, small, tile):
    """dp_conv3x3s2_fwd (round 5: conv2 of the first bottleneck of stages 2-4, 3x3 / stride 2 / pad 1, de-interleaved LDS
    image) against F.conv2d at 1e-5 of the output scale; one-hot weights exact (padding above / left of every image, the
    seams between the images a tile spans, no padding below / right); bit-identical to the even pixels of the stride-1
    kernel's result (same summation order); with the GroupNorm fold bit-identical to normalising first.  Round 6: with
    448-, 128- and 64-pixel tiles (the small ones start mid-row and cross image boundaries anywhere) and the launcher's pick."""
    import os
    if DEV == "cpu" and not small and not (os.environ.get("DORPATCH_EMU_FULL", "0") == "1" and C <= 16):
        pytest.skip("through the fibre emulation this case takes minutes: GPU (or DORPATCH_EMU_FULL=1 for the narrow ones)")
    if DEV == "cpu" and tile in (1, 3) and not os.environ.get("DORPATCH_EMU_FULL", "0") == "1":
        pytest.skip("emulation: the launcher's pick and the 64-pixel tile cover the new geometry; all tiles run on the GPU")
    ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, tile << 4)      # bits 4-6: forced pixel tile (0: the launcher's rule)
    try:
        _conv3x3s2_case(N, C, O, S, small)
    finally:
        ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 0)


def _conv3x3s2_case(N, C, O, S, small):
    g = torch.Generator().manual_seed(C + S)
    x = torch.randn(N, C, S, S, generator=g)
    x[0] += torch.arange(float(S)).view(1, S, 1) * 0.1 + torch.arange(float(S)).view(1, 1, S) * 0.01
    w = torch.randn(O, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    want = F.conv2d(x, w, stride=2, padding=1)
    xd, wt = x.to(DEV).contiguous(), ops.pack_conv3x3_weights(w).to(DEV)
    got = ops.conv3x3s2_fwd(xd, wt)
    assert tuple(got.shape) == (N, O, S // 2, S // 2)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    w1 = torch.zeros(O, C, 3, 3)
    for o in range(O):
        w1[o, (5 * o + 3) % C, o % 3, (o // 3) % 3] = 1.0
    got1 = ops.conv3x3s2_fwd(xd, ops.pack_conv3x3_weights(w1).to(DEV)).cpu()
    assert torch.equal(got1, F.conv2d(x, w1, stride=2, padding=1))
    if DEV != "cpu" or small:        # the stride-1 kernel on the same input: its even pixels, bit for bit
        full = ops.conv3x3_fwd(xd, wt)
        assert torch.equal(got, full[:, :, ::2, ::2])
    if C % 32 == 0 or C == 16:       # GroupNorm fold (groups of 32, or of 16 channels in the emulation-sized case)
        G = 32 if C % 32 == 0 else 16
        xr = (x * 1.5 + 0.3).to(DEV).contiguous()
        gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
        beta = (torch.randn(C, generator=g) * 0.2 + 1.0).to(DEV)
        y, mean, rstd, _ = ops.gn_relu_fwd(xr, gamma, beta, G, 1e-5)
        _, _, ab, _ = ops.gn_stats(xr, gamma, beta, G, 1e-5)
        assert torch.equal(ops.conv3x3s2_fwd(xr, wt, ab=ab), ops.conv3x3s2_fwd(y, wt))


CONV1X1_CASES = [   # (N, C, O, H, emulation-sized)
    (3, 32, 64, 14, True),      # 1.3 tiles of 2.3 planes each, ragged last tile, 2 K-chunks (the double buffer)
    (11, 16, 128, 7, True),     # flat mode: 9 whole images per tile + a ragged second tile, two output-channel groups, ONE chunk
    (1, 16, 64, 28, True),      # 1.75 tiles inside one plane
    (2, 64, 256, 56, False), (2, 256, 64, 56, False), (2, 256, 128, 56, False), (3, 128, 512, 28, False),
    (3, 512, 128, 28, False), (3, 512, 256, 28, False), (5, 256, 1024, 14, False), (5, 1024, 256, 14, False),
    (5, 1024, 512, 14, False), (19, 512, 2048, 7, False), (19, 2048, 512, 7, False), (10, 1024, 2048, 7, False),
    (1, 256, 128, 96, False), (2, 512, 2048, 12, False), (1, 128, 512, 48, False),      # planes of a 384 x 384 input
]


CONV1X1_TILES = {0: "auto", 1: "448px", 2: "256px", 3: "128px", 4: "64px"}     # bits 4-6 of DP_DEBUG_CONV1X1_VARIANT


@pytest.mark.parametrize("tile", sorted(CONV1X1_TILES), ids=lambda t: CONV1X1_TILES[t])
@pytest.mark.parametrize("N,C,O,H,small", CONV1X1_CASES)
def test_conv1x1_on_the_matrix_cores_matches_conv2d(N, C, O, H, small, tile):
    """dp_conv1x1_fwd (round 5: the backbone's 1x1 / 1 convolutions as an NCHW-in-place GEMM on v_mfma_f32_32x32x2_f32)
    against F.conv2d, forward and — the same entry point on the transposed weights — input gradient: exact-f32
    arithmetic, another summation order -> 1e-5 of the output scale; one-hot weights (output channel o copies input channel
    (5 o + 3) % C) must be EXACT for every pixel of every image, incl. the seams between the images a tile spans and the
    ragged last tile; `res` (the epilogue add) exact against the sum; in place (out = res) as well.  Round 6: with every
    pixel tile the launcher can pick and with its own choice."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    if DEV == "cpu" and tile in (2, 3):
        pytest.skip("emulation: 448 / 64 pixels + the launcher's pick here, every tile in test_conv1x1_pixel_tiles_are_bit_identical")
    ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, tile << 4)       # bits 4-6: forced pixel tile (0: the launcher's rule)
    try:
        _conv1x1_case(N, C, O, H)
    finally:
        ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, 0)


def _conv1x1_case(N, C, O, H):
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, H, generator=g)
    x[0] += torch.arange(float(H)).view(1, H, 1) * 0.1 + torch.arange(float(H)).view(1, 1, H) * 0.01
    w = torch.randn(O, C, 1, 1, generator=g) / C ** 0.5
    want = F.conv2d(x, w)
    xd = x.to(DEV).contiguous()
    wt = ops.pack_conv1x1_weights(w).to(DEV)
    got = ops.conv1x1_fwd(xd, wt)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))
    w1 = torch.zeros(O, C, 1, 1)
    for o in range(O):
        w1[o, (5 * o + 3) % C] = 1.0
    got1 = ops.conv1x1_fwd(xd, ops.pack_conv1x1_weights(w1).to(DEV)).cpu()
    assert torch.equal(got1, F.conv2d(x, w1))
    # epilogue add: exactly (conv + res) as the kernel alone computes conv; and in place
    res = torch.randn(N, O, H, H, generator=g)
    got_r = ops.conv1x1_fwd(xd, wt, res=res.to(DEV))
    assert torch.equal(got_r.cpu(), got.cpu() + res)
    acc = res.to(DEV).clone()
    ops.conv1x1_fwd(xd, wt, res=acc, out=acc)
    assert torch.equal(acc.cpu(), got_r.cpu())
    # input gradient = the same kernel on dy with the transposed weights
    if C % 64 == 0:
        dy = torch.randn(N, O, H, H, generator=g)
        want_dx = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                      (True, False, False))[0]
        got_dx = ops.conv1x1_fwd(dy.to(DEV).contiguous(), ops.pack_conv1x1_weights(w, transpose=True).to(DEV)).cpu()
        np.testing.assert_allclose(got_dx.numpy(), want_dx.numpy(), rtol=0, atol=1e-5 * float(want_dx.abs().max()))


def test_conv1x1_launch_variants_are_bit_identical():
    """dp_debug_set(DP_DEBUG_CONV1X1_VARIANT): the workgroup-id maps, non-temporal stores and the placement of the LDS
    staging are A/B knobs for measurements — every combination must produce the same bits (tile count not a multiple of 8, three
    output-channel groups, 2 K-chunks)."""
    from dorpatch_amd import _lib
    g = torch.Generator().manual_seed(7)
    x = torch.randn(5, 32, 14, 14, generator=g).to(DEV)
    wt = ops.pack_conv1x1_weights(torch.randn(192, 32, 1, 1, generator=g)).to(DEV)
    try:
        outs = []
        for variant in (0, 1, 2, 4, 8, 14):
            ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, variant)
            outs.append(ops.conv1x1_fwd(x, wt).cpu())
    finally:
        ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, 0)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("C,O,H", [(64, 256, 56), (512, 128, 28), (1024, 256, 14), (2048, 512, 7)])
def test_conv1x1_at_the_bench_size_agrees_with_the_library(C, O, H):
    """BASELINE configs[1]'s micro-batch (N = 512) on four of ResNetV2-50's 1x1 shapes (row mode at 56 / 28 / 14, whole-image
    tiles at 7; K from 64 to 2048): the hand-written kernel against the library's convolution on the same device (an
    independent implementation) to 3e-5 of the output scale; the LAST image against F.conv2d on the CPU; the epilogue add
    against a separate add, bit for bit; same bits twice."""
    if DEV == "cpu":
        pytest.skip("bench-sized: GPU only")
    N = 512
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(O, C, 1, 1, generator=g) / C ** 0.5
    xd, wt = x.to(DEV), ops.pack_conv1x1_weights(w).to(DEV)
    got = ops.conv1x1_fwd(xd, wt)
    ref = F.conv2d(xd, w.to(DEV))
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 3e-5 * scale
    np.testing.assert_allclose(got[-1:].cpu().numpy(), F.conv2d(x[-1:], w).numpy(), rtol=0, atol=3e-5 * scale)
    assert torch.equal(got, ops.conv1x1_fwd(xd, wt))
    res = torch.randn(N, O, H, H, generator=g).to(DEV)
    assert torch.equal(ops.conv1x1_fwd(xd, wt, res=res), got + res)


@pytest.mark.parametrize("C,S", [(128, 56), (256, 28), (512, 14)])
def test_conv3x3_stride2_pair_at_the_bench_size_agrees_with_the_library(C, S):
    """BASELINE configs[1]'s micro-batch (N = 512) on ResNetV2-50's three stride-2 3x3 convolutions: dp_conv3x3s2_fwd and the
    input gradient dp_conv3x3s2_bwd against the library on the same device (3e-5 of the scale), the last image against the CPU,
    same bits twice."""
    if DEV == "cpu":
        pytest.skip("bench-sized: GPU only")
    N = 512
    g = torch.Generator().manual_seed(C + S)
    x = torch.randn(N, C, S, S, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    xd, wdv = x.to(DEV), w.to(DEV)
    got = ops.conv3x3s2_fwd(xd, ops.pack_conv3x3_weights(w).to(DEV))
    ref = F.conv2d(xd, wdv, stride=2, padding=1)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 3e-5 * scale
    np.testing.assert_allclose(got[-1:].cpu().numpy(), F.conv2d(x[-1:], w, stride=2, padding=1).numpy(), rtol=0, atol=3e-5 * scale)
    assert torch.equal(got, ops.conv3x3s2_fwd(xd, ops.pack_conv3x3_weights(w).to(DEV)))
    dy = torch.randn(N, C, S // 2, S // 2, generator=g).to(DEV)
    wb = ops.pack_conv3x3s2_dgrad_weights(w).to(DEV)
    got_dx = ops.conv3x3s2_bwd(dy, wb, C)
    ref_dx = torch.ops.aten.convolution_backward(dy, xd, wdv, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))[0]
    sdx = float(ref_dx.abs().max())
    assert float((got_dx - ref_dx).abs().max()) <= 3e-5 * sdx
    assert torch.equal(got_dx, ops.conv3x3s2_bwd(dy, wb, C))


def test_stem_convolution_and_its_input_gradient_at_the_bench_size_agree_with_the_library():
    """N = 512 images of 224 x 224 through dp_stem_conv_fwd and dp_stem_dgrad against the library on the same device."""
    if DEV == "cpu":
        pytest.skip("bench-sized: GPU only")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(512, 3, 224, 224, generator=g).to(DEV)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12.0)
    wdv = w.to(DEV)
    got = ops.stem_conv_fwd(x, ops.pack_stem_weights(w).to(DEV))
    ref = F.conv2d(x, wdv, stride=2, padding=3)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(got, ops.stem_conv_fwd(x, ops.pack_stem_weights(w).to(DEV)))
    dy = torch.randn(512, 64, 112, 112, generator=g).to(DEV)
    got_dx = ops.stem_dgrad(dy, wdv)
    ref_dx = torch.ops.aten.convolution_backward(dy, x, wdv, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (True, False, False))[0]
    assert float((got_dx - ref_dx).abs().max()) <= 3e-5 * float(ref_dx.abs().max())


CONV1X1_TILE_CASES = [   # (N, C, O, H, emulation-sized)
    (3, 32, 64, 14, True),      # row mode: tiles of 448 / 256 / 128 / 64 pixels cut 588 pixels at different seams
    (11, 16, 128, 7, True),     # flat mode: 9 / 5 / 2 / 1 whole images per tile
    (9, 64, 256, 56, False), (7, 1024, 256, 14, False), (23, 2048, 512, 7, False), (5, 128, 512, 28, False),
]


@pytest.mark.parametrize("N,C,O,H,small", CONV1X1_TILE_CASES)
def test_conv1x1_pixel_tiles_are_bit_identical(N, C, O, H, small):
    """Round 6: the pixel tile of dp_conv1x1_fwd is a scheduling choice of the launcher — an output
    element's k-walk (channels ascending, one fmaf chain) is the same in every tile, so plain, folded (GroupNorm-apply in
    the staging) and residual-add launches must give THE SAME BITS with 448-, 256-, 128- and 64-pixel tiles and with the
    launcher's own pick; the output starts as NaN: every element is written by every tile."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.randn(N, C, H, H, generator=g).to(DEV)
    wt = ops.pack_conv1x1_weights(torch.randn(O, C, 1, 1, generator=g) / C ** 0.5).to(DEV)
    res = torch.randn(N, O, H, H, generator=g).to(DEV)
    fold = (H * H) % 4 == 0
    if fold:
        ab = torch.randn(N, C, 2, generator=g).to(DEV)
    outs = {}
    try:
        for tile in sorted(CONV1X1_TILES):
            ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, tile << 4)
            out = torch.full((N, O, H, H), float("nan"), device=DEV)
            got = [ops.conv1x1_fwd(x, wt, out=out).cpu(), ops.conv1x1_fwd(x, wt, res=res).cpu()]
            if fold:
                got.append(ops.conv1x1_fwd(x, wt, ab=ab).cpu())
                got.append(ops.conv1x1_fwd(x, wt, ab=ab, res=res).cpu())
            outs[tile] = got
    finally:
        ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, 0)
    assert not torch.isnan(outs[1][0]).any()
    for tile, got in outs.items():
        for a, b in zip(got, outs[1]):
            assert torch.equal(a, b), "tile %s differs from the 448-pixel tile" % CONV1X1_TILES[tile]


CONV1X1_FOLD_CASES = [   # (N, C, O, H, emulation-sized)
    (3, 32, 64, 14, True), (1, 64, 64, 28, True),
    (2, 256, 64, 56, False), (3, 128, 512, 28, False), (5, 1024, 256, 14, False), (1, 256, 128, 96, False),
]


@pytest.mark.parametrize("N,C,O,H,small", CONV1X1_FOLD_CASES)
def test_conv1x1_with_folded_groupnorm_is_bit_identical_to_the_two_kernels(N, C, O, H, small):
    """dp_gn_stats + dp_conv1x1_fwd(ab) — GroupNorm-apply + ReLU while staging the operand — must equal
    dp_gn_relu_fwd followed by the plain dp_conv1x1_fwd BIT FOR BIT (same statistics, same x * a + b, same MFMA walk), with
    and without the residual add in front of the norm; mean / rstd / the sum are the GroupNorm kernel's own."""
    if DEV == "cpu" and not small:
        pytest.skip("through the fibre emulation this case takes minutes: GPU only")
    G = 32
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, C, H, H, generator=g) * 1.5 + 0.3).to(DEV)
    r = torch.randn(N, C, H, H, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.2).to(DEV)
    wt = ops.pack_conv1x1_weights(torch.randn(O, C, 1, 1, generator=g) / C ** 0.5).to(DEV)
    for res in (None, r):
        y, mean, rstd, s = ops.gn_relu_fwd(x, gamma, beta, G, 1e-5, res=res)
        want = ops.conv1x1_fwd(y, wt)
        mean2, rstd2, ab, s2 = ops.gn_stats(x, gamma, beta, G, 1e-5, res=res)
        assert torch.equal(mean2, mean) and torch.equal(rstd2, rstd) and torch.equal(s2, s)
        got = ops.conv1x1_fwd(s2, wt, ab=ab)
        assert torch.equal(got, want)


# ---------------------------------------------------------------- fused GroupNorm + ReLU (backbone, a-8)
GN_SHAPES = [  # (N, C, H, W): every (V, T) register variant, the HW = 49 per-lane channel path, streaming
    (3, 64, 56, 56),      # L4 = 1568  -> <4,512>
    (2, 256, 56, 56),     # L4 = 6272  -> <7,1024>
    (2, 128, 56, 56),     # L4 = 3136  -> <4,1024>
    (3, 128, 28, 28),     # L4 = 784   -> <4,256>
    (2, 256, 14, 14),     # L4 = 392   -> <2,256>
    (5, 512, 7, 7),       # L4 = 196   -> <1,256>, HW % 4 != 0
    (2, 2048, 7, 7),      # L4 = 784, HW % 4 != 0
    (1, 256, 96, 96),     # L4 = 18432 -> fwd <18,1024>, bwd k_gn_relu_bwd_big<18,0,9,3>: dh in registers, half of xh in LDS (384x384 inputs)
    (1, 128, 96, 96),     # L4 = 9216  -> fwd <9,1024>, bwd k_gn_relu_bwd_big<9,9,0,3>: all in registers
    (2, 512, 48, 48),     # L4 = 9216  -> the same, 16 channels per group
    (1, 256, 88, 88),     # L4 = 15488 -> the V = 18 kernels with a ragged tail (lanes past the group)
    (1, 512, 44, 44),     # L4 = 7744  -> the V = 9 kernels with a ragged tail
    (1, 256, 94, 94),     # L4 = 17672, HW % 4 == 0; (1, 256, 93, 93) below: HW % 4 != 0 -> streaming kernels
    (1, 256, 93, 93),
    (2, 32, 4, 4),        # one channel per group, tiny
]


@pytest.mark.parametrize("shape", GN_SHAPES)
def test_gn_relu_matches_torch(shape):
    """y = relu(group_norm(x)) vs torch (float64 on the CPU; the op is third-party maths — timm's
    GroupNormAct — so the reference is torch, not the oracle) and the input gradient vs the analytic
    GroupNorm backward in float64.  The ReLU gate is a step function: where |z| < 1e-5 fp32 and fp64
    may legitimately disagree, so the gradient reference takes the gate from the kernel's own output
    and the gate itself is checked everywhere else."""
    N, C, H, W = shape
    G = 32
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    dy = torch.randn(N, C, H, W, generator=g)
    y, mean, rstd, _ = ops.gn_relu_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), G, 1e-5)
    dx = ops.gn_relu_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), beta.to(DEV), mean, rstd, G)
    y, dx = y.cpu(), dx.cpu()
    xd = x.double()
    z = torch.nn.functional.group_norm(xd, G, gamma.double(), beta.double(), 1e-5)
    np.testing.assert_allclose(y.numpy(), torch.relu(z).float().numpy(), rtol=2e-5, atol=2e-6)
    xg = xd.view(N, G, -1)
    mu, var = xg.mean(-1, keepdim=True), xg.var(-1, unbiased=False, keepdim=True)
    r = (var + 1e-5).rsqrt()
    np.testing.assert_allclose(mean.cpu().numpy(), mu.reshape(-1).float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), r.reshape(-1).float().numpy(), rtol=1e-5)
    gate = y > 0
    assert bool(((gate == (z > 0)) | (z.abs() < 1e-5)).all())
    xh = ((xg - mu) * r)
    dxh = (dy.double() * gate * gamma.double().view(1, C, 1, 1)).view(N, G, -1)
    want = r * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
    np.testing.assert_allclose(dx.numpy(), want.view(N, C, H, W).float().numpy(), rtol=1e-4, atol=2e-5)
    # and the analytic form above IS torch's autograd (checked in float64, gate from z)
    xr = xd.clone().requires_grad_(True)
    (dxr,) = torch.autograd.grad(torch.relu(torch.nn.functional.group_norm(xr, G, gamma.double(), beta.double(), 1e-5)),
                                 xr, dy.double())
    dxh2 = (dy.double() * (z > 0) * gamma.double().view(1, C, 1, 1)).view(N, G, -1)
    want2 = r * (dxh2 - dxh2.mean(-1, keepdim=True) - xh * (dxh2 * xh).mean(-1, keepdim=True))
    np.testing.assert_allclose(want2.view(N, C, H, W).numpy(), dxr.numpy(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("C,H", [(256, 56), (1024, 14)])
def test_gn_relu_at_the_bench_size_agrees_with_the_library(C, H):
    """N = 512 samples (BASELINE configs[1]'s micro-batch) through dp_gn_relu_fwd / dp_gn_stats / dp_gn_relu_bwd: forward and
    input gradient against torch's own GroupNorm + ReLU (autograd) on the same device, the statistics pass against the full
    kernel bit for bit, the last sample against float64 on the CPU."""
    if DEV == "cpu":
        pytest.skip("bench-sized: GPU only")
    N, G = 512, 32
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, C, H, H, generator=g) * 1.5 + 0.3).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    dy = torch.randn(N, C, H, H, generator=g).to(DEV)
    y, mean, rstd, _ = ops.gn_relu_fwd(x, gamma, beta, G, 1e-5)
    m2, r2, ab, _ = ops.gn_stats(x, gamma, beta, G, 1e-5)
    assert torch.equal(mean, m2) and torch.equal(rstd, r2)
    dx = ops.gn_relu_bwd(dy, x, gamma, beta, mean, rstd, G)
    xr = x.clone().requires_grad_(True)
    z = F.group_norm(xr, G, gamma, beta, 1e-5)
    ref = torch.relu(z).detach()
    assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # the gate of the reference is the kernel's own (|z| < 1e-5: fp32 implementations may legitimately disagree on it)
    (want,) = torch.autograd.grad(z, xr, dy * (y > 0))
    assert float((dx - want).abs().max()) <= 1e-4 * float(want.abs().max())
    z64 = F.group_norm(x[-1:].cpu().double(), G, gamma.cpu().double(), beta.cpu().double(), 1e-5)
    np.testing.assert_allclose(y[-1:].cpu().numpy(), torch.relu(z64).float().numpy(), rtol=2e-5, atol=2e-6)


def test_gn_relu_autograd_function_in_module():
    """GroupNormAct routes frozen GPU inputs through the fused kernels; trainable / CPU stay on torch."""
    from dorpatch_amd.resnetv2 import GroupNormAct
    m = GroupNormAct(64).to(DEV)
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5)
        m.bias.normal_(0, 0.2)
    x = torch.randn(4, 64, 28, 28, device=DEV)
    xa = x.clone().requires_grad_(True)
    ya = m(xa)                                  # trainable affine params -> torch path
    assert "GnRelu" not in type(ya.grad_fn).__name__
    (ga,) = torch.autograd.grad(ya.sum() + (ya * ya).sum(), xa)
    for p in m.parameters():
        p.requires_grad_(False)
    xb = x.clone().requires_grad_(True)
    yb = m(xb)
    assert "GnRelu" in type(yb.grad_fn).__name__
    (gb,) = torch.autograd.grad(yb.sum() + (yb * yb).sum(), xb)
    np.testing.assert_allclose(yb.detach().cpu().numpy(), ya.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    bad = (gb - ga).abs() > 2e-4 * ga.abs() + 2e-5          # a ReLU gate at |z| ~ ulp may flip one group
    assert float(bad.float().mean()) < 1e-2
    GroupNormAct.fused = False
    try:
        assert "GnRelu" not in type(m(xb).grad_fn).__name__
    finally:
        GroupNormAct.fused = True


def test_resnetv2_fused_equals_unfused():
    """Whole frozen ResNetV2-50x1-BiT: logits and input gradient with the fused GN+ReLU / pooling / stem kernels vs the
    eager composition (same library convolutions either way), on the WELL-CONDITIONED seeded weights
    (resnetv2.seeded_init_: ReLU gate flips under re-ordered fp32 sums are rare), so the bound is tight — 1e-4 of the
    scale with at most 0.1 % of the elements beyond it (one flipped gate).  The fp64-oracle statement of the same
    network is tests/test_backbone_parity_gpu.py; this one isolates the hand-written backbone kernels."""
    from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, GroupNormAct, resnetv2_50x1_bit, seeded_init_
    net = seeded_init_(resnetv2_50x1_bit(1000), gn_bias=WELL_CONDITIONED_GN_BIAS).fold_weight_standardization().freeze().to(DEV)
    n, side = (4, 224) if DEV != "cpu" else (2, 96)     # the CPU emulation re-runs this test on a smaller problem
    x = torch.rand(n, 3, side, side, generator=torch.Generator().manual_seed(2)).to(DEV) * 2 - 1
    dl = torch.randn(n, 1000, generator=torch.Generator().manual_seed(3)).to(DEV)
    outs = []
    for fused in (True, False):
        GroupNormAct.fused = fused
        try:
            xi = x.clone().requires_grad_(True)
            lg = net(xi)
            (gx,) = torch.autograd.grad(lg, xi, dl)
            outs.append((lg.detach().cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64)))
        finally:
            GroupNormAct.fused = True
    for got, want in zip(outs[0], outs[1]):
        err = np.abs(got - want) / np.abs(want).max()
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 5e-4 and (err > 1e-4).mean() <= 1e-3, \
            (np.linalg.norm(got - want) / np.linalg.norm(want), err.max(), (err > 1e-4).mean())


@pytest.mark.parametrize("shape", [(2, 256, 56, 56), (3, 512, 28, 28), (2, 2048, 7, 7), (1, 256, 96, 96), (1, 512, 48, 48),
                                   (1, 256, 88, 88)])
def test_add_gn_relu_fusion(shape):
    """Residual add fused into GroupNorm+ReLU: (x, res) -> (x + res, relu(gn(x + res))) and the backward
    with the shortcut gradient folded in, against the unfused kernels (bit-exact: same arithmetic)
    and through autograd against the eager torch composition."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H)
    x, r = torch.randn(N, C, H, W, generator=g).to(DEV), torch.randn(N, C, H, W, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    dy, ds = torch.randn(N, C, H, W, generator=g).to(DEV), torch.randn(N, C, H, W, generator=g).to(DEV)
    y, mean, rstd, s = ops.gn_relu_fwd(x, gamma, beta, 32, 1e-5, res=r)
    assert torch.equal(s, x + r)
    y0, mean0, rstd0, s0 = ops.gn_relu_fwd(s, gamma, beta, 32, 1e-5)
    assert s0 is s and torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
    dx = ops.gn_relu_bwd(dy, s, gamma, beta, mean, rstd, 32, dres=ds)
    dx0 = ops.gn_relu_bwd(dy, s, gamma, beta, mean, rstd, 32)
    assert torch.equal(dx, dx0 + ds)
    # autograd: both outputs used, only y used, only s used
    xa, ra = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    sa, ya = ops.AddGnReluFunction.apply(xa, ra, gamma, beta, 32, 1e-5)
    gx, gr = torch.autograd.grad([sa, ya], [xa, ra], [ds, dy])
    assert torch.equal(gx, dx) and torch.equal(gr, dx)
    sa, ya = ops.AddGnReluFunction.apply(xa, ra, gamma, beta, 32, 1e-5)
    (gx,) = torch.autograd.grad(ya, xa, dy)
    assert torch.equal(gx, dx0)
    sa, ya = ops.AddGnReluFunction.apply(xa, ra, gamma, beta, 32, 1e-5)
    (gx,) = torch.autograd.grad(sa, xa, ds)
    assert torch.equal(gx, ds)
    # eager torch on the GPU
    xe, re_ = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    se = xe + re_
    ye = torch.relu(torch.nn.functional.group_norm(se, 32, gamma, beta, 1e-5))
    ge, _ = torch.autograd.grad([se, ye], [xe, re_], [ds, dy])
    np.testing.assert_allclose(y.cpu().numpy(), ye.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    bad = (ge - dx).abs() > 2e-4 * ge.abs() + 2e-5
    assert float(bad.float().mean()) < 1e-2


@pytest.mark.parametrize("shape", [(3, 64, 112, 112), (2, 5, 8, 16), (1, 64, 192, 192)])
def test_pad_maxpool_matches_torch(shape):
    """Fused ConstantPad2d(1,0)+MaxPool2d(3,2) (BiT stem) vs eager torch: forward exact, backward exact
    (pure routing), including all-negative border windows where the padded zero wins and ties."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, C, H, W, generator=g)
    x[0, 0, :3, :] = -1.0 - torch.rand(3, W, generator=g)        # top windows: pad zero is the max
    x[0, 0, :, :3] = -1.0 - torch.rand(H, 3, generator=g)        # left windows likewise
    x[0, 1 % C, 4:8, 4:8] = 0.5                                   # ties inside windows
    x = x.to(DEV)
    dy = torch.randn(N, C, H // 2, W // 2, generator=g).to(DEV)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(torch.nn.functional.pad(xr, (1, 1, 1, 1), value=0.0), 3, 2, 0)
    (gr,) = torch.autograd.grad(yr, xr, dy)
    y, code = ops.pad_maxpool_fwd(x)
    assert torch.equal(y, yr.detach())
    assert int(code.max()) <= 8
    gx = ops.pad_maxpool_bwd(dy, code, H, W)
    # an input pixel can win up to 4 windows: torch's scatter adds them in atomic order, the gather here
    # in a fixed order -> equal up to the rounding of a 3- or 4-term fp32 sum
    np.testing.assert_allclose(gx.cpu().numpy(), gr.cpu().numpy(), rtol=1e-6, atol=1e-6)
    assert torch.equal(gx == 0, gr == 0)                            # identical routing
    xa = x.clone().requires_grad_(True)
    (ga,) = torch.autograd.grad(ops.PadMaxPoolFunction.apply(xa), xa, dy)
    assert torch.equal(ga, gx)


@pytest.mark.parametrize("N,K,H", [(3, 64, 224), (2, 8, 40), (1, 64, 384), (2, 5, 34)])
def test_stem_dgrad_matches_torch(N, K, H):
    """dp_stem_dgrad vs torch's conv2d input gradient (float64 on the CPU): direct fp32 gather with
    one rounding per MAC — tolerance 2e-5 of the gradient scale (K*49-term dot products)."""
    g = torch.Generator().manual_seed(K + H)
    w = torch.randn(K, 3, 7, 7, generator=g) * 0.1
    dy = torch.randn(N, K, H // 2, H // 2, generator=g)
    x = torch.zeros(N, 3, H, H, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(torch.nn.functional.conv2d(x, w.double(), None, 2, 3), x, dy.double())
    got = ops.stem_dgrad(dy.to(DEV), w.to(DEV)).cpu().double()
    scale = float(want.abs().max())
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=2e-5 * scale)
    # through autograd, frozen weight
    xa = torch.rand(N, 3, H, H, generator=g).to(DEV).requires_grad_(True)
    (ga,) = torch.autograd.grad(ops.StemConvFunction.apply(xa, w.to(DEV)), xa, dy.to(DEV))
    assert torch.equal(ga.cpu().double(), got)


@pytest.mark.parametrize("shape", [(2, 8, 56, 56), (3, 5, 28, 28), (2, 4, 14, 14), (1, 3, 6, 10)])
def test_subsample2_and_its_accumulating_adjoint(shape):
    """dp_subsample2 / dp_subsample2_add vs torch strided slicing (exact: pure data movement + one add)."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(N, C, H, W, generator=g).to(DEV)
    y = ops.subsample2(x)
    assert torch.equal(y, x[:, :, ::2, ::2])
    base = torch.randn(N, C, H, W, generator=g).to(DEV)
    dy = torch.randn(N, C, H // 2, W // 2, generator=g).to(DEV)
    want = base.clone()
    want[:, :, ::2, ::2] += dy
    got = ops.subsample2_add_(base.clone(), dy)
    assert torch.equal(got, want)


@pytest.mark.parametrize("stride,N,C,mid,O,H", [(2, 3, 16, 8, 32, 28), (1, 2, 8, 8, 32, 12), (2, 2, 12, 4, 24, 14)])
def test_dual_conv1x1_matches_two_convolutions(stride, N, C, mid, O, H):
    """conv1 + (strided) downsample of the same pre-activation as one autograd node: outputs and the summed
    input gradient equal F.conv2d + autograd (fp32 dot products in another order: rtol 1e-5 of the scale);
    unused outputs are handled (None gradients)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(stride * 100 + H)
    pre = torch.randn(N, C, H, H, generator=g).to(DEV)
    w1 = (torch.randn(mid, C, 1, 1, generator=g) / C ** 0.5).to(DEV)
    wd = (torch.randn(O, C, 1, 1, generator=g) / C ** 0.5).to(DEV)
    Ho = H // stride
    d1, dd = torch.randn(N, mid, H, H, generator=g).to(DEV), torch.randn(N, O, Ho, Ho, generator=g).to(DEV)
    pr = pre.clone().requires_grad_(True)
    b_ref, s_ref = F.conv2d(pr, w1), F.conv2d(pr, wd, stride=stride)
    (g_ref,) = torch.autograd.grad([b_ref, s_ref], pr, [d1, dd])
    pa = pre.clone().requires_grad_(True)
    b, s = ops.DualConv1x1Function.apply(pa, w1, wd, stride)
    (g_got,) = torch.autograd.grad([b, s], pa, [d1, dd])
    for got, want in ((b, b_ref), (s, s_ref), (g_got, g_ref)):
        assert got.shape == want.shape
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5,
                                   atol=1e-5 * float(want.detach().abs().max()))
    # only one of the two outputs used downstream
    b, s = ops.DualConv1x1Function.apply(pa, w1, wd, stride)
    (g_b,) = torch.autograd.grad(b, pa, d1)
    b, s = ops.DualConv1x1Function.apply(pa, w1, wd, stride)
    (g_s,) = torch.autograd.grad(s, pa, dd)
    np.testing.assert_allclose((g_b + g_s).cpu().numpy(), g_ref.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(g_ref.abs().max()))


@pytest.mark.parametrize("B,S,K,H,dual", [(2, 8, 64, 224, False), (1, 5, 8, 40, True), (3, 4, 5, 36, False), (1, 16, 8, 56, False)])
def test_stem_dgrad_reduce_equals_stem_dgrad_then_apply_bwd(B, S, K, H, dual):
    """dp_stem_dgrad_reduce == dp_apply_bwd(dp_stem_dgrad(.)) bit for bit (same arithmetic, same slab partition),
    with and without the fused 1/std, single and dual masks, accumulate, sizes that do not fill the tiles."""
    if DEV == "cpu" and K * H >= 64 * 224:          # the CPU emulation re-runs this on a smaller problem
        K, S = 8, 3
    g = torch.Generator().manual_seed(B * 100 + S + H)
    w = (torch.randn(K, 3, 7, 7, generator=g) * 0.1).to(DEV)
    dy = torch.randn(B * S, K, H // 2, H // 2, generator=g).to(DEV)
    table_np = masks.universe_rects(H, 2)
    table = ops.upload_table(table_np, DEV)
    rng = np.random.RandomState(S)
    idx = torch.from_numpy(np.stack([rng.choice(len(table_np), S, replace=False) for _ in range(B)])).int().to(DEV)
    idx2 = torch.from_numpy(np.stack([rng.choice(len(table_np), S, replace=False) for _ in range(B)])).int().to(DEV) if dual else None
    for norm in (ops.RAW_NORM, ops.make_norm([0.5, 0.4, 0.3], [0.5, 0.25, 0.2], 0.5)):
        G = ops.stem_dgrad(dy, w)
        want = ops.apply_bwd(G, table, idx, idx2, norm, B=B)
        got = ops.stem_dgrad_reduce(dy, w, table, idx, idx2, norm, B=B)
        assert torch.equal(got, want)
        base = torch.randn(B, 3, H, H, generator=g).to(DEV)
        want_acc = ops.apply_bwd(G, table, idx, idx2, norm, B=B, out=base.clone(), accumulate=True)
        got_acc = ops.stem_dgrad_reduce(dy, w, table, idx, idx2, norm, B=B, out=base.clone(), accumulate=True)
        assert torch.equal(got_acc, want_acc)


def test_launch_stream_handle_is_torchs_current_stream():
    """ops._stream() takes the raw handle from torch's C entry points (no Stream object per launch): it must be the very
    stream torch.cuda.current_stream() names — on the default stream, inside a side-stream context, and back outside —
    and a kernel launched through ops inside the context is ordered with the torch ops of that stream."""
    if DEV == "cpu":
        pytest.skip("the emulation has no streams (tests/hipemu/patch.py hands the kernels a null handle)")

    def both():
        return ops._stream().value or 0, torch.cuda.current_stream().cuda_stream or 0
    a, b = both()
    assert a == b
    x, p, m = _rand(2, 3, 56, 56, seed=1), _rand(2, 3, 56, 56, seed=2), _rand(2, 1, 56, 56, seed=3)
    want = R.clip(m, p, x, 4.0) + x
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        a, b = both()
        assert a == b == side.cuda_stream
        xs = [t.to(DEV, non_blocking=True) * 1.0 for t in (m, p, x)]      # produced ON the side stream
        adv = ops.blend(xs[0], xs[1], xs[2], 4.0)[0]
        got = adv.cpu()
    torch.cuda.current_stream().wait_stream(side)
    a, b = both()
    assert a == b
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
