"""Occlusion-mask geometry as rectangle tables.

The reference materialises every PatchCleanser mask as a ``(1, H, W)`` bool
tensor (``defenses/PatchCleanser.py:44-59``) and the EOT universe as a
``(2520, 1, H, W)`` bool tensor — 126 MB at 224², 372 MB at 384²
(``attack.py:83-85``).  Here a mask is a short list of axis-aligned windows
``(row0, row1, col0, col1)`` (half-open); the HIP kernels test pixels against
the windows on the fly, so the hot path reads **zero** mask bytes from HBM.

Table layout (host numpy / device int32): ``(n_mask, R, 4)``; unused window
slots are the empty rectangle ``(0, 0, 0, 0)``.
"""
import math

import numpy as np
import torch

NUM_MASK_PER_AXIS = 6                       # PatchCleanser.py:13
DROPOUT_SIZES = (0.015, 0.03, 0.06, 0.12)   # attack.py:83


def window_params(img_size, patch_ratio=0.03, n_patch=1):
    """(mask_size, stride, window_size) — PatchCleanser.py:11-16."""
    mask_size = math.floor(math.sqrt(img_size ** 2 * patch_ratio / n_patch))
    stride = int(np.ceil((img_size - mask_size + 1) / NUM_MASK_PER_AXIS))
    window_size = mask_size + stride - 1
    return mask_size, stride, window_size


def single_rects(img_size, patch_ratio=0.03, n_patch=1):
    """(36, 1, 4) int32: window ``6*i + j`` covers rows ``[stride*i, +window)``,
    cols ``[stride*j, +window)`` clipped to the image (PatchCleanser.py:51-58)."""
    _, stride, window = window_params(img_size, patch_ratio, n_patch)
    n = NUM_MASK_PER_AXIS
    out = np.zeros((n * n, 1, 4), dtype=np.int32)
    for i in range(n):
        for j in range(n):
            r0, c0 = stride * i, stride * j
            out[i * n + j, 0] = (r0, min(img_size, r0 + window), c0, min(img_size, c0 + window))
    return out


def pair_index():
    """All window pairs ``a < b`` in the order the reference keeps them
    (upper triangle, row-major — PatchCleanser.py:23-29)."""
    n = NUM_MASK_PER_AXIS ** 2
    a, b = np.triu_indices(n, k=1)
    return a.astype(np.int64), b.astype(np.int64)


def double_rects(img_size, patch_ratio=0.03, n_patch=1):
    """(630, 2, 4) int32: entry k is the union of windows ``a_k`` and ``b_k``."""
    single = single_rects(img_size, patch_ratio, n_patch)[:, 0]
    a, b = pair_index()
    return np.stack([single[a], single[b]], axis=1).astype(np.int32)


def mask_set_rects(img_size, dropout_size, dropout):
    """Rect-table twin of ``attack.get_mask_set`` (attack.py:25-31)."""
    if dropout == 1:
        return single_rects(img_size, dropout_size)
    if dropout == 2:
        return double_rects(img_size, dropout_size)
    raise ValueError("dropout must be 1 or 2 (the reference's dropout=0 path builds no mask set)")


def universe_rects(img_size, dropout=2, dropout_sizes=DROPOUT_SIZES):
    """The EOT mask universe: the sets of all four dropout sizes concatenated
    (attack.py:83-85).  Universe index k -> (ratio k // n_per, entry k % n_per)."""
    tables = [mask_set_rects(img_size, ds, dropout) for ds in dropout_sizes]
    return np.concatenate(tables, axis=0)


def pad_rects(table, R):
    """Pad the window axis of a table to R slots with empty rectangles."""
    n, r, _ = table.shape
    if r == R:
        return table
    out = np.zeros((n, R, 4), dtype=np.int32)
    out[:, :r] = table
    return out


def rects_to_bool(table, H, W=None, device="cpu"):
    """Materialise a table as the reference's ``(n, 1, H, W)`` bool masks
    (True = pixel kept, False = occluded).  Compatibility / tests only."""
    W = H if W is None else W
    table = np.asarray(table)
    rows = torch.arange(H, device=device).view(1, 1, H, 1)
    cols = torch.arange(W, device=device).view(1, 1, 1, W)
    t = torch.as_tensor(table, device=device).long()
    occluded = torch.zeros((table.shape[0], 1, H, W), dtype=torch.bool, device=device)
    for r in range(table.shape[1]):
        r0, r1, c0, c1 = (t[:, r, k].view(-1, 1, 1, 1) for k in range(4))
        occluded |= (rows >= r0) & (rows < r1) & (cols >= c0) & (cols < c1)
    return ~occluded


def bool_to_rects(masks, max_rects=4):
    """The reference's ``(n, 1, H, W)`` (or ``(n, H, W)``) bool masks (True = pixel kept) -> an ``(n, R, 4)`` rectangle
    table with EXACTLY the same occluded pixels, R <= ``max_rects``.

    Each mask's occluded region is cut into horizontal bands of identical rows and every band into its column runs —
    the union of two axis-aligned windows (every PatchCleanser mask, ``PatchCleanser.py:44-59``) needs at most 4
    rectangles this way (3 when the windows overlap in their columns).  Raises ``ValueError`` for a mask that needs more
    than ``max_rects`` rectangles: such a mask is not a PatchCleanser mask and is refused rather than approximated."""
    m = masks.detach().cpu().numpy() if isinstance(masks, torch.Tensor) else np.asarray(masks)
    if m.ndim == 4:
        if m.shape[1] != 1:
            raise ValueError("bool masks must be (n, 1, H, W) or (n, H, W), got %s" % (m.shape,))
        m = m[:, 0]
    if m.ndim != 3 or m.dtype != np.bool_:
        raise ValueError("bool masks must be a bool array of shape (n, 1, H, W) or (n, H, W)")
    n, H, W = m.shape
    per_mask, R = [], 1
    for k in range(n):
        occ = ~m[k]
        rects = []
        rows = np.flatnonzero(occ.any(axis=1))
        if rows.size:
            r_lo, r_hi = int(rows[0]), int(rows[-1]) + 1
            body = occ[r_lo:r_hi]
            cuts = np.flatnonzero((body[1:] != body[:-1]).any(axis=1)) + 1          # first row of every new band
            edges = np.concatenate([[0], cuts, [r_hi - r_lo]])
            for b0, b1 in zip(edges[:-1], edges[1:]):
                line = np.concatenate([[0], body[b0].astype(np.int8), [0]])
                d = np.diff(line)
                for c0, c1 in zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1)):
                    rects.append((r_lo + int(b0), r_lo + int(b1), int(c0), int(c1)))
        if len(rects) > max_rects:
            raise ValueError("mask %d needs %d rectangles (> %d): not a union of two axis-aligned windows"
                             % (k, len(rects), max_rects))
        R = max(R, len(rects))
        per_mask.append(rects)
    out = np.zeros((n, R, 4), dtype=np.int32)
    for k, rects in enumerate(per_mask):
        for r, rect in enumerate(rects):
            out[k, r] = rect
    return out
