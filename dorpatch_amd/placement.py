"""EXTENSION — random affine placement of the patch per EOT sample.

Not part of the reference: CGCL-codes/DorPatch blends the patch at identity placement only
(``attack.py:184-185``).  ``BASELINE.json``'s ``north_star`` names "random affine placement ×
random occlusion masks", so the fused apply kernels take an optional per-sample 2x3 affine; with
``placement=None`` (the default everywhere) nothing here runs and the path is the reference's.

Semantics (pinned by ``oracle/restatement.warp_delta`` = ``F.affine_grid`` + ``F.grid_sample`` and
``tests/test_placement*.py``): EOT sample (b, s) sees

    occlude( x[b] + warp(delta[b], theta[b, s]) ),      delta = mask * (pattern - x) * scale   (utils.clip)

``theta`` maps OUTPUT pixel coordinates to SOURCE pixel coordinates of ``delta`` (bilinear, zeros outside).
The structural / density / group-lasso terms, the failure sweep and PatchCleanser keep acting on the
un-warped ``adv_x = x + delta`` exactly as in the reference; only the EOT forward/backward sees the warp, and
``d loss / d delta`` is the exact adjoint (``dp_apply_affine_bwd``).
"""
import numpy as np


class RandomAffine(object):
    """Rotation (degrees), isotropic scale and translation (pixels) about the image centre, drawn per (image,
    sample) from a legacy ``np.random.RandomState`` — one generator per image, so sample-sharded ranks that start
    from identical states draw identical placements."""

    def __init__(self, max_rotate_deg=10.0, scale=(0.9, 1.1), max_translate_px=8.0):
        self.max_rotate_deg = float(max_rotate_deg)
        self.scale = (float(scale[0]), float(scale[1]))
        self.max_translate_px = float(max_translate_px)

    def draw(self, rng, S, H, W):
        """(S, 2, 3) float32: output -> source pixel maps."""
        rot = np.deg2rad(rng.uniform(-self.max_rotate_deg, self.max_rotate_deg, size=S))
        sc = rng.uniform(self.scale[0], self.scale[1], size=S)
        tx = rng.uniform(-self.max_translate_px, self.max_translate_px, size=S)
        ty = rng.uniform(-self.max_translate_px, self.max_translate_px, size=S)
        return compose(rot, sc, tx, ty, H, W)


def compose(rot, scale, tx, ty, H, W):
    """Output->source maps of "rotate by rot, scale by `scale`, shift by (tx, ty)" about the centre: the patch
    content at source position q appears at output position  R_s (q - c) + c + t, so source = R_s^-1 (o - c - t) + c."""
    rot, scale, tx, ty = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (rot, scale, tx, ty))
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    cos, sin = np.cos(rot) / scale, np.sin(rot) / scale          # R_s^-1 = (1/scale) * R(-rot)
    a00, a01, a10, a11 = cos, sin, -sin, cos
    t0 = cx - (a00 * (cx + tx) + a01 * (cy + ty))
    t1 = cy - (a10 * (cx + tx) + a11 * (cy + ty))
    return np.stack([np.stack([a00, a01, t0], -1), np.stack([a10, a11, t1], -1)], -2).astype(np.float32)


def identity(S):
    return np.tile(np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float32), (S, 1, 1))


def invert(theta):
    """(…, 2, 3) -> inverse affine maps, float64 arithmetic, float32 result."""
    t = np.asarray(theta, dtype=np.float64)
    a, b, c, d = t[..., 0, 0], t[..., 0, 1], t[..., 1, 0], t[..., 1, 1]
    det = a * d - b * c
    ia, ib, ic, id_ = d / det, -b / det, -c / det, a / det
    t0 = -(ia * t[..., 0, 2] + ib * t[..., 1, 2])
    t1 = -(ic * t[..., 0, 2] + id_ * t[..., 1, 2])
    return np.stack([np.stack([ia, ib, t0], -1), np.stack([ic, id_, t1], -1)], -2).astype(np.float32)


def to_normalized(theta, H, W):
    """Pixel-coordinate output->source maps -> the ``theta`` of ``F.affine_grid(..., align_corners=False)`` for a
    square image (H == W): with u = (2 p + 1) / W - 1,  u_src = A u_out + [A 1 (1 - 1/W) + (2 t + 1) / W - 1]."""
    assert H == W, "normalised form implemented for square images"
    t = np.asarray(theta, dtype=np.float64)
    A = t[..., :, :2]
    off = A.sum(-1) * (1.0 - 1.0 / W) + (2.0 * t[..., :, 2] + 1.0) / W - 1.0
    return np.concatenate([A, off[..., None]], -1).astype(np.float32)
