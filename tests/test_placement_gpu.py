"""EXTENSION tests — per-sample affine placement of the patch (dorpatch_amd/placement.py, dp_apply_affine_fwd/_bwd).

Nothing in the reference to compare with (it places the patch at identity only), so the oracle is
``oracle/restatement.warp_delta`` = ``F.affine_grid`` + ``F.grid_sample`` (bilinear, zeros, align_corners=False) and
its autograd, and the anchor to the reference is: identity placement reproduces the reference path bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dorpatch_amd import masks, ops  # noqa: E402
from dorpatch_amd import placement as PL  # noqa: E402
from dorpatch_amd.attack import DorPatch, HotLoop  # noqa: E402
from oracle import restatement as R  # noqa: E402
from oracle import toy_models  # noqa: E402

DEV = "cuda:0"
NORM = ([0.5, 0.4, 0.3], [0.5, 0.25, 0.2])


def _setup(B, S, H, seed=0, dual=False, affine=(25.0, (0.7, 1.3), 6.0)):
    g = torch.Generator().manual_seed(seed)
    x, delta = torch.rand(B, 3, H, H, generator=g), (torch.rand(B, 3, H, H, generator=g) - 0.5) * 0.3
    table_np = masks.universe_rects(H, 2)
    rng = np.random.RandomState(seed)
    idx_np = np.stack([rng.choice(len(table_np), S, replace=False) for _ in range(B)])
    idx2_np = np.stack([rng.choice(len(table_np), S, replace=False) for _ in range(B)]) if dual else None
    theta = np.stack([PL.RandomAffine(*affine).draw(rng, S, H, H) for _ in range(B)])
    return x, delta, table_np, idx_np, idx2_np, theta


def _keep(table_np, idx_np, H):
    keep = masks.rects_to_bool(table_np, H)                  # (n,1,H,W) bool
    return torch.stack([keep[torch.from_numpy(r)] for r in idx_np])      # (B,S,1,H,W)


def test_identity_placement_is_the_reference_path_bit_for_bit():
    B, S, H = 2, 6, 56
    x, delta, table_np, idx_np, idx2_np, _ = _setup(B, S, H, dual=True)
    table = ops.upload_table(table_np, DEV)
    idx, idx2 = torch.from_numpy(idx_np).int().to(DEV), torch.from_numpy(idx2_np).int().to(DEV)
    theta = torch.from_numpy(np.stack([PL.identity(S)] * B)).to(DEV)
    norm = ops.make_norm(*NORM, 0.5)
    xd, dd = x.to(DEV), delta.to(DEV)
    got = ops.apply_affine_fwd(xd, dd, theta, table, idx, idx2, norm)
    want = ops.apply_fwd((xd + dd).contiguous(), table, idx, idx2, norm)
    assert torch.equal(got, want)
    G = torch.randn(B * S, 3, H, H, generator=torch.Generator().manual_seed(1)).to(DEV)
    gb = ops.apply_affine_bwd(G, theta, theta.clone(), table, idx, idx2, norm, B=B)
    assert torch.equal(gb, ops.apply_bwd(G, table, idx, idx2, norm, B=B))


# The tiled kernels stage a 32 x 32 tile's footprint in LDS when it fits (<= 64 wide, <= 48 x 48 pixels) and fall back to
# per-pixel global gathers otherwise; the cases below reach both paths of both kernels:
#   default placement range (rotation 10 deg, scale 0.9-1.1)      forward staged, backward staged
#   the wide range of round 2 at 56 / 40 px (ragged tiles)         mixed, depending on the draw
#   scale ~0.45 (patch shown 0.45x: source footprint 2.2x the tile) forward SLOW path, backward staged
#   scale ~2.2  (output region of a source tile 2.9x the tile)      forward staged, backward SLOW path
@pytest.mark.parametrize("B,S,H,affine", [(2, 5, 56, (25.0, (0.7, 1.3), 6.0)), (1, 3, 40, (25.0, (0.7, 1.3), 6.0)),
                                          (1, 4, 96, (10.0, (0.9, 1.1), 8.0)), (1, 3, 96, (20.0, (0.42, 0.48), 3.0)),
                                          (1, 3, 96, (20.0, (2.1, 2.3), 3.0)), (1, 2, 72, (180.0, (0.9, 1.1), 2.0))])
def test_affine_apply_matches_grid_sample_oracle_and_its_adjoint(B, S, H, affine):
    x, delta, table_np, idx_np, _, theta = _setup(B, S, H, seed=3, affine=affine)
    table = ops.upload_table(table_np, DEV)
    idx = torch.from_numpy(idx_np).int().to(DEV)
    norm = ops.make_norm(*NORM, 0.5)
    th = torch.from_numpy(theta).to(DEV)
    got = ops.apply_affine_fwd(x.to(DEV), delta.to(DEV), th, table, idx, None, norm).view(B, S, 3, H, H).cpu()
    # oracle: F.affine_grid + F.grid_sample, then occlusion (fill 0.5) and normalisation as the reference does them
    keep = _keep(table_np, idx_np, H)
    dl = delta.clone().requires_grad_(True)
    placed = x[:, None] + R.warp_delta(dl, torch.from_numpy(PL.to_normalized(theta, H, H)))
    masked = placed * keep + 0.5 * ~keep
    mean, std = torch.tensor(NORM[0]).view(1, 1, 3, 1, 1), torch.tensor(NORM[1]).view(1, 1, 3, 1, 1)
    want = (masked - mean) / std
    # grid_sample works in normalised coordinates: its fp32 tap positions differ from the kernel's pixel-space ones by
    # ~1e-5 px, which a patch shown at 0.45x (neighbouring outputs 2.2 source pixels apart) turns into 2e-5 of the value
    tol = 2e-5 if min(affine[1]) >= 0.6 else 5e-5
    np.testing.assert_allclose(got.numpy(), want.detach().numpy(), rtol=0, atol=tol)
    # backward: d/d delta of <out, G>
    G = torch.randn(B, S, 3, H, H, generator=torch.Generator().manual_seed(5))
    (want_g,) = torch.autograd.grad((want * G).sum(), dl)
    got_g = ops.apply_affine_bwd(G.view(B * S, 3, H, H).to(DEV), th, torch.from_numpy(PL.invert(theta)).to(DEV), table,
                                 idx, None, norm, B=B).cpu()
    np.testing.assert_allclose(got_g.numpy(), want_g.numpy(), rtol=0, atol=tol * float(want_g.abs().max()))
    # exact adjoint (bilinear weights computed by the same expression in both kernels): <A d, G> == <d, A^T G>
    d2 = torch.randn(B, 3, H, H, generator=torch.Generator().manual_seed(6))
    zero_x = torch.zeros_like(x)
    raw = ops.RAW_NORM
    Ad = ops.apply_affine_fwd(zero_x.to(DEV), d2.to(DEV), th, table, idx, None, ops.make_norm(None, None, 0.0))
    AtG = ops.apply_affine_bwd(G.view(B * S, 3, H, H).to(DEV), th, torch.from_numpy(PL.invert(theta)).to(DEV), table, idx,
                               None, raw, B=B)
    lhs = float((Ad.double().cpu().view(-1) * G.double().view(-1)).sum())
    rhs = float((d2.double().view(-1) * AtG.double().cpu().view(-1)).sum())
    assert abs(lhs - rhs) <= 1e-5 * (abs(lhs) + 1.0), (lhs, rhs)


@pytest.mark.parametrize("B,S,H,affine", [(2, 5, 56, (25.0, (0.7, 1.3), 6.0)), (1, 7, 96, (20.0, (0.42, 1.2), 3.0))])
def test_forward_is_bit_identical_for_any_number_of_samples_per_workgroup(B, S, H, affine, monkeypatch):
    """The forward walks `samples per workgroup` samples with the next footprint in flight; small problems get 1 from the
    launcher, so the walk is forced here (dp_debug_set): 2, 3 (ragged last chunk), 8 (> S: one chunk) against 1.
    The second case mixes staged and slow-path samples (scale 0.42 .. 1.2) inside one chunk."""
    x, delta, table_np, idx_np, idx2_np, theta = _setup(B, S, H, seed=11, dual=True, affine=affine)
    table = ops.upload_table(table_np, DEV)
    idx, idx2 = torch.from_numpy(idx_np).int().to(DEV), torch.from_numpy(idx2_np).int().to(DEV)
    norm = ops.make_norm(*NORM, 0.5)
    th = torch.from_numpy(theta).to(DEV)
    outs = {}
    from dorpatch_amd._lib import DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK as KNOB
    try:
        for spb in (1, 2, 3, 8):
            ops.debug_set(KNOB, spb)
            outs[spb] = ops.apply_affine_fwd(x.to(DEV), delta.to(DEV), th, table, idx, idx2, norm).cpu()
    finally:
        ops.debug_set(KNOB, 0)
    for spb in (2, 3, 8):
        assert torch.equal(outs[spb], outs[1]), spb


@pytest.mark.parametrize("B,S,H,affine", [(2, 5, 56, (10.0, (0.9, 1.1), 8.0)), (2, 5, 56, (25.0, (0.7, 1.3), 6.0)),
                                          (1, 6, 96, (30.0, (0.45, 0.6), 3.0)), (1, 4, 224, (10.0, (0.9, 1.1), 8.0))])
def test_backward_gather_variants_are_bit_identical(B, S, H, affine):
    """dp_apply_affine_bwd's gather (one branch per candidate) against the branch-free A/B variant (DP_DEBUG_AFFINE_GATHER
    = 1: a row's records, weights and gradients requested back to back, folded in with selects — measured slower, kept
    as evidence): the same candidates in the same order, so the same bits — default placement range, the wide range (windows of up to
    4 x 4 half-widths -> more than one column group per row), small scales (regions that go to the per-pixel path) and
    the full 224 x 224 plane with two mask sets."""
    from dorpatch_amd._lib import DP_DEBUG_AFFINE_GATHER as KNOB
    x, delta, table_np, idx_np, idx2_np, theta = _setup(B, S, H, seed=31, dual=True, affine=affine)
    table = ops.upload_table(table_np, DEV)
    idx, idx2 = torch.from_numpy(idx_np).int().to(DEV), torch.from_numpy(idx2_np).int().to(DEV)
    th, thi = torch.from_numpy(theta).to(DEV), torch.from_numpy(PL.invert(theta)).to(DEV)
    G = torch.randn(B * S, 3, H, H, generator=torch.Generator().manual_seed(9)).to(DEV)
    norm = ops.make_norm(*NORM, 0.5)
    outs = []
    try:
        for variant in (1, 2, 0):      # 2: hit compaction (round 5: a 64-bit mask of the window's taps, then the lane's own hits)
            ops.debug_set(KNOB, variant)
            outs.append(ops.apply_affine_bwd(G, th, thi, table, idx, idx2, norm, B=B).cpu())
    finally:
        ops.debug_set(KNOB, 0)
    assert torch.equal(outs[0], outs[2]), float((outs[0] - outs[2]).abs().max())
    assert torch.equal(outs[1], outs[2]), float((outs[1] - outs[2]).abs().max())
    assert outs[0].abs().max() > 0


def test_full_size_launch_adjoint_and_walk_invariance():
    """The benchmark's own launch — 64 images x 32 samples at 224^2, default placement range, PatchCleanser double masks:
    the launcher's choice (8 samples per forward workgroup) is bit-identical to a walk of 1; <A d, G> == <d, A^T G> in
    fp64 over all 2048 samples (size-independent property: no CPU oracle at this size); a second pass of either kernel
    gives the same bits.  GPU only (2.5 GB of samples)."""
    if DEV == "cpu":
        pytest.skip("2048 x 3 x 224 x 224 samples: GPU only")
    from dorpatch_amd._lib import DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK as KNOB
    B, S, H = 64, 32, 224
    x, delta, table_np, idx_np, idx2_np, theta = _setup(B, S, H, seed=21, dual=True, affine=(10.0, (0.9, 1.1), 8.0))
    table = ops.upload_table(table_np, DEV)
    idx, idx2 = torch.from_numpy(idx_np).int().to(DEV), torch.from_numpy(idx2_np).int().to(DEV)
    th, thi = torch.from_numpy(theta).to(DEV), torch.from_numpy(PL.invert(theta)).to(DEV)
    raw = ops.make_norm(None, None, 0.0)
    zero_x, dd = torch.zeros_like(x).to(DEV), delta.to(DEV)
    Ad = ops.apply_affine_fwd(zero_x, dd, th, table, idx, idx2, raw)
    assert torch.equal(Ad, ops.apply_affine_fwd(zero_x, dd, th, table, idx, idx2, raw))
    ops.debug_set(KNOB, 1)
    try:
        assert torch.equal(Ad, ops.apply_affine_fwd(zero_x, dd, th, table, idx, idx2, raw))
    finally:
        ops.debug_set(KNOB, 0)
    G = torch.randn(B * S, 3, H, H, generator=torch.Generator(device=DEV).manual_seed(7), device=DEV)
    AtG = ops.apply_affine_bwd(G, th, thi, table, idx, idx2, ops.RAW_NORM, B=B)
    assert torch.equal(AtG, ops.apply_affine_bwd(G, th, thi, table, idx, idx2, ops.RAW_NORM, B=B))
    lhs = float((Ad.double().view(-1) * G.double().view(-1)).sum())
    rhs = float((dd.double().view(-1) * AtG.double().view(-1)).sum())
    print("full-size adjoint: <A d, G> = %.9e, <d, A^T G> = %.9e" % (lhs, rhs))
    assert abs(lhs - rhs) <= 1e-5 * (abs(lhs) + 1.0), (lhs, rhs)


def test_size_limits_are_rejected_before_any_launch():
    """include/dorpatch_hip.h: 12 * H * W < 2^31 and H, W < 2^22 for the placement entry points (32-bit byte offsets,
    24-bit multiplies in the kernels' addressing).  The calls claim such sizes over tiny buffers: they must return
    hipErrorInvalidValue (1) without launching."""
    import ctypes
    lib = ops._lib.load()
    buf = torch.zeros(256, device=DEV)
    theta = torch.from_numpy(PL.identity(1)).to(DEV)
    table = ops.upload_table(np.zeros((1, 1, 4), dtype=np.int32), DEV)
    idx = torch.zeros(1, dtype=torch.int32, device=DEV)
    norm = ops.make_norm(None, None, 0.5)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    fwd = lambda H, W: lib.dp_apply_affine_fwd(p(buf), p(buf), p(theta), p(table), 1, p(idx), None, 1, 1, 1, H, W,
                                               ctypes.byref(norm), p(buf), None)
    bwd = lambda H, W: lib.dp_apply_affine_bwd(p(buf), p(theta), p(theta), p(table), 1, p(idx), None, 1, 1, 1, H, W,
                                               ctypes.byref(norm), p(buf), None)
    assert fwd(4, 4) == 0 and bwd(4, 4) == 0                      # the same call at a legal size goes through
    torch.cuda.synchronize() if DEV != "cpu" else None
    assert fwd(16384, 16384) == 1                                 # 12 * H * W = 3.2e9
    assert fwd(4, 1 << 22) == 1 and bwd(4, 1 << 22) == 1 and bwd(1 << 22, 4) == 1


def test_backward_slab_of_more_than_64_samples():
    """dp_apply_affine_bwd takes its per-sample uniforms from one lane per sample, 64 samples at a time; a slab holds more
    than 64 samples only when tiles x B >= 4096 and S > 128 (one slab).  84 images x 132 samples at 224^2, identity
    placement: bit-identical to dp_apply_bwd (GPU only: 6.7 GB of gradients)."""
    if DEV == "cpu":
        pytest.skip("6.7 GB problem: GPU only")
    B, S, H = 84, 132, 224
    assert ops._lib.load().dp_apply_bwd_nslab(B, S, H * H) == 1
    table_np = masks.universe_rects(H, 2)
    rng = np.random.RandomState(5)
    idx = torch.from_numpy(np.stack([rng.choice(len(table_np), S, replace=False) for _ in range(B)])).int().to(DEV)
    table = ops.upload_table(table_np, DEV)
    theta = torch.from_numpy(np.stack([PL.identity(S)] * B)).to(DEV)
    norm = ops.make_norm(*NORM, 0.5)
    G = torch.randn(B * S, 3, H, H, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV)
    got = ops.apply_affine_bwd(G, theta, theta.clone(), table, idx, None, norm, B=B)
    assert torch.equal(got, ops.apply_bwd(G, table, idx, None, norm, B=B))


class FixedPlacement(object):
    def __init__(self, theta):
        self.theta = theta
        self.k = 0

    def draw(self, rng, S, H, W):
        out = self.theta[self.k]
        self.k += 1
        return out


class FixedDraw(object):
    def __init__(self, rows):
        self.rows = list(rows)

    def choice(self, a, n, replace=False):
        return np.asarray(self.rows.pop(0)).copy()


@pytest.mark.parametrize("stage", [0, 1])
def test_hot_loop_step_with_placement_matches_oracle(stage):
    """One HotLoop.step with a placement: losses and the parameter gradients against the oracle step whose EOT
    samples see x + grid_sample(delta) (autograd through the warp); B = 2 images, each with its own draws."""
    H, S, B = 56, 6, 2
    model = toy_models.NormModel(toy_models.make_toy(gain=2.0), toy_models.Normalize()).to(DEV)
    g = torch.Generator().manual_seed(17)
    x, m0, p0 = torch.rand(B, 3, H, H, generator=g), torch.rand(B, 1, H, H, generator=g), torch.rand(B, 3, H, H, generator=g)
    if stage == 1:
        m0 = (m0 > 0.8).float()
    y = torch.tensor([3, 5])
    rng = np.random.RandomState(2)
    rows = [rng.choice(2520, S, replace=False) for _ in range(B)]
    theta = np.stack([PL.RandomAffine(15.0, (0.85, 1.2), 4.0).draw(rng, S, H, H) for _ in range(B)])
    got = {}
    hook = lambda d: got.update({k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in d.items()})
    loop = HotLoop(DorPatch(micro_batch=8, verbose=False), model, x.to(DEV), 0.12, 10, "t/cfg/sub", 0, y.to(DEV), True,
                   1e-2, 1e-1, 0, 1, 10, 7, 'topk', 2, S, 1e-3, 1e-3, 4.0, False,
                   dict(init_mask=m0, init_pattern=p0, rngs=[FixedDraw([rows[b]]) for b in range(B)],
                        failure_refresh=10 ** 9, step_hook=hook, placement=FixedPlacement(list(theta))))
    loop.stage = stage
    loop.step(1)
    loop.close()
    assert np.array_equal(got["theta"], theta)
    uni = R.mask_universe(H, 2)
    keep = torch.stack([uni[torch.from_numpy(r)] for r in rows])                       # (B,S,1,H,W)
    cpu = toy_models.NormModel(toy_models.make_toy(gain=2.0), toy_models.Normalize())
    want = R.eot_step(cpu, x, m0, p0, y, keep, stage=stage, targeted=True, n_classes=10, lr=0.01,
                      theta_norm=torch.from_numpy(PL.to_normalized(theta, H, H)))
    np.testing.assert_allclose(got["loss_adv"], want["loss_adv"].numpy(), rtol=1e-4, atol=1e-5)
    gw = want["grad_pattern"].numpy()
    np.testing.assert_allclose(got["grad_pattern"].numpy(), gw, rtol=1e-3, atol=1e-3 * np.abs(gw).max())
    if stage == 0:
        gm, gmw = got["grad_mask"].numpy(), want["grad_mask"].numpy()
        np.testing.assert_allclose(gm, gmw, rtol=1e-3, atol=1e-3 * np.abs(gmw).max())
