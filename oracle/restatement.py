"""TEST INFRASTRUCTURE — CPU restatement of the DorPatch EOT hot path.

NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` import this module, as the checker / the
reported CPU baseline.  It restates, with plain ``torch`` CPU ops and autograd
used exactly the way the reference uses it, what ``/root/reference`` computes on
the hot path; every function cites the reference lines it follows.  It is pinned
against the committed ``tests/golden/*.npz`` fixtures, which ``oracle/gen_golden.py``
records by executing the *unmodified* reference through ``oracle/ref_shim.py`` in the build
container (``tests/test_oracle_golden.py``, ``tests/test_patchcleanser_oracle.py``; the end-metric fixtures are
consumed by ``tests/test_end_metric_gpu.py`` / ``tests/test_end_metric_emu.py``), and, in the build container, checked
bit for bit against the unmodified reference executed in place (``tests/test_vs_live_reference.py``).

Parity status: pinned for everything below; the timm backbone is opaque to this
path (any ``nn.Module``) and is unpinned (see oracle/__init__.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DROPOUT_SIZES = (0.015, 0.03, 0.06, 0.12)      # attack.py:83
N_AXIS = 6                                     # PatchCleanser.py:13


# ----------------------------------------------------------------------------- a-10
def window_geometry(img_size, patch_ratio, n_patch=1):
    """PatchCleanser.py:11-16 -> (mask_size, stride, window_size)."""
    mask_size = math.floor(math.sqrt(img_size * img_size * patch_ratio / n_patch))
    stride = int(np.ceil((img_size - mask_size + 1) / N_AXIS))
    return mask_size, stride, mask_size + stride - 1


def single_masks(img_size, patch_ratio):
    """PatchCleanser.py:44-59: 36 bool masks (True = keep), window 6*i+j at rows stride*i.., cols stride*j.."""
    _, stride, window = window_geometry(img_size, patch_ratio)
    keep = torch.ones((N_AXIS * N_AXIS, 1, img_size, img_size), dtype=torch.bool)
    for k in range(N_AXIS * N_AXIS):
        i, j = divmod(k, N_AXIS)
        keep[k, 0, stride * i: min(img_size, stride * i + window),
             stride * j: min(img_size, stride * j + window)] = False
    return keep


def double_masks(img_size, patch_ratio):
    """PatchCleanser.py:23-29: products of all single-mask pairs a < b (upper triangle, row-major)."""
    single = single_masks(img_size, patch_ratio)
    n = single.shape[0]
    pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    a_idx = torch.tensor([p[0] for p in pairs])
    b_idx = torch.tensor([p[1] for p in pairs])
    return single[a_idx] & single[b_idx]


def mask_universe(img_size, dropout=2):
    """attack.py:25-31, 83-85: the sets of the four dropout sizes, concatenated."""
    build = {1: single_masks, 2: double_masks}[dropout]
    return torch.cat([build(img_size, r) for r in DROPOUT_SIZES], dim=0)


# ----------------------------------------------------------------------------- a-2
def clip(mask, pattern, x, eps):
    """utils.py:105-110: blend by the mask, then L2-rescale with a detached factor."""
    delta = mask * (pattern - x)
    norm = delta.flatten(1).norm(p=2, dim=1).detach()
    factor = (eps / norm).clamp(max=1.0).view(-1, 1, 1, 1)
    return delta * factor


# ----------------------------------------------------------------------------- a-7
def cw_loss(logits, y, n_classes, targeted, confidence):
    """attack.py:16-23."""
    onehot = F.one_hot(y, n_classes)
    real = (logits * onehot).sum(1)
    other = ((1.0 - onehot) * logits - onehot * 1e4).max(1)[0]
    margin = (confidence + other - real) if targeted else (confidence + real - other)
    return margin.clamp(min=0.0)


# ----------------------------------------------------------------------------- a-5
def local_variance(x):
    """attack.py:33-39.  The minuend is a *detached clone*; only the subtracted
    neighbour carries gradient; last column / last row keep the raw pixel."""
    lr_ = x.clone().detach()
    lr_[:, :, :, :-1].sub_(x[:, :, :, 1:]).abs_()
    ud_ = x.clone().detach()
    ud_[:, :, :-1, :].sub_(x[:, :, 1:, :]).abs_()
    return lr_ + ud_, lr_, ud_


def min_var_weighted_variance(x):
    """attack.py:41-45."""
    total, lr_, ud_ = local_variance(x)
    return total * torch.where(lr_ > ud_, ud_, lr_)


def struct_loss(adv_x, local_var_x):
    """attack.py:227-228."""
    return torch.mean(min_var_weighted_variance(adv_x).mean(1) / (local_var_x + 1e-5), (1, 2))


# ----------------------------------------------------------------------------- a-6
def density_loss(mask):
    """attack.py:77-80, 237: unbiased variance of the (W//8)-window sums of the mask."""
    win = int(mask.shape[-1] // 8)
    sums = F.conv2d(mask, torch.ones(1, 1, win, win, dtype=mask.dtype), stride=win)
    return sums.flatten(1).var(1)


def group_lasso(mask, unit=7):
    """attack.py:72-74, 243-244."""
    cell = F.conv2d(mask ** 2, torch.ones(1, 1, unit, unit, dtype=mask.dtype), stride=unit)
    return unit * cell.sqrt().sum((1, 2, 3))


# ----------------------------------------------------------------------------- a-4
def occlude(adv_x, keep):
    """attack.py:206 / PatchCleanser.py:99-100: img * mask + 0.5 * ~mask.
    adv_x (B,3,H,W), keep (S,1,H,W) bool -> (B,S,3,H,W)."""
    return adv_x[:, None] * keep + 0.5 * ~keep


# ----------------------------------------------------------------------------- extension: affine placement
def warp_delta(delta, theta_norm):
    """EXTENSION oracle (nothing in the reference to follow — SURVEY §0): delta (B,3,H,W) placed under S affine maps per
    image, theta_norm (B,S,2,3) in ``F.affine_grid`` convention -> (B,S,3,H,W).  bilinear, zeros outside,
    align_corners=False."""
    B, S = theta_norm.shape[:2]
    _, C, H, W = delta.shape
    grid = F.affine_grid(theta_norm.reshape(B * S, 2, 3).to(delta.dtype), (B * S, C, H, W), align_corners=False)
    rep = delta[:, None].expand(B, S, C, H, W).reshape(B * S, C, H, W)
    return F.grid_sample(rep, grid, mode="bilinear", padding_mode="zeros", align_corners=False).view(B, S, C, H, W)


# ----------------------------------------------------------------------------- one step
def eot_step(model, x, mask, pattern, y, keep, *, stage, targeted, n_classes, confidence=0.1,
             structured=1e-3, density=1e-3, coeff_group_lasso=1e-5, eps=4.0, lr=None,
             clip_min=0.0, clip_max=1.0, unit=7, keep_dual=None, local_var_x=None, theta_norm=None):
    """One pass of attack.py:184-247 (+ the update of 333-342 when ``lr`` is given) for
    B >= 1 images treated as independent problems that share the sampled masks ``keep``
    (S,1,H,W) — or per-image masks when ``keep`` is (B,S,1,H,W).

    Returns a dict of every intermediate the parity tests compare.  ``mask`` / ``pattern``
    are not modified; updated copies are returned under ``new_mask`` / ``new_pattern``.
    ``structured`` / ``coeff_group_lasso`` / ``lr`` may be floats or per-image sequences."""
    B = x.shape[0]
    mask = mask.detach().clone().requires_grad_(stage == 0)
    pattern = pattern.detach().clone().requires_grad_(True)
    if local_var_x is None:
        local_var_x = local_variance(x)[0].mean(1)                          # attack.py:100
    delta = clip(mask, pattern, x, eps)                                     # :184
    adv_x = delta + x                                                       # :185
    if theta_norm is not None:              # EXTENSION (placement.py): each sample sees x + warp(delta, theta[b,s])
        placed = x[:, None] + warp_delta(delta, theta_norm)
        kk = keep if keep.dim() == 5 else keep[None]
        masked = placed * kk + 0.5 * ~kk
        if keep_dual is not None:
            masked = masked * keep_dual + 0.5 * ~keep_dual
    elif keep.dim() == 4:
        masked = occlude(adv_x, keep)                                       # :206
        if keep_dual is not None:
            masked = masked * keep_dual + 0.5 * ~keep_dual                  # :218
    else:
        masked = adv_x[:, None] * keep + 0.5 * ~keep
        if keep_dual is not None:
            masked = masked * keep_dual + 0.5 * ~keep_dual
    S = masked.shape[1]
    logits = model(masked.reshape((-1,) + masked.shape[2:]))                # :220-222
    y_rep = y.view(B, 1).expand(B, S).reshape(-1)                           # :98
    tflags = np.broadcast_to(np.asarray(targeted, dtype=bool), (B,))
    rows = []
    for b in range(B):
        rows.append(cw_loss(logits[b * S:(b + 1) * S], y_rep[b * S:(b + 1) * S], n_classes,
                            bool(tflags[b]), confidence))
    loss_adv = torch.stack(rows)                                            # :224-225
    loss_struc = struct_loss(adv_x, local_var_x)                            # :227-228
    coef_s = torch.as_tensor(np.broadcast_to(np.asarray(structured, dtype=np.float64), (B,)).copy(),
                             dtype=x.dtype)
    loss = loss_adv.mean(1) + coef_s * loss_struc                           # :230-233
    out = dict(adv_x=adv_x.detach(), logits=logits.detach(), loss_adv=loss_adv.detach(),
               loss_struc=loss_struc.detach(), scale=None)
    if stage == 0:
        dens = density_loss(mask)                                           # :237
        gl = group_lasso(mask, unit)                                        # :243-244
        coef_g = torch.as_tensor(np.broadcast_to(np.asarray(coeff_group_lasso, dtype=np.float64), (B,)).copy(),
                                 dtype=x.dtype)
        loss = loss + density * dens + coef_g * gl                          # :239-245
        out.update(density=dens.detach(), group_lasso=gl.detach())
    loss.sum().backward()                                                   # :247
    out["loss"] = loss.detach()
    out["grad_pattern"] = pattern.grad.detach().clone()
    out["grad_mask"] = mask.grad.detach().clone() if stage == 0 else torch.zeros_like(mask)
    if lr is not None:                                                      # :333-342
        lr_t = torch.as_tensor(np.broadcast_to(np.asarray(lr, dtype=np.float32), (B,)).copy()).to(x.dtype).view(B, 1, 1, 1)
        with torch.no_grad():
            new_pattern = (pattern - lr_t * pattern.grad.sign()).clamp(clip_min, clip_max)
            new_mask = (mask - lr_t * mask.grad.sign()).clamp(clip_min, clip_max) if stage == 0 else mask.detach()
        out["new_pattern"], out["new_mask"] = new_pattern.detach(), new_mask.detach()
    return out


# ----------------------------------------------------------------------------- next-3
def patch_selection(mask, patch_budget, unit=7):
    """attack.py:363-382 (selection='topk')."""
    B = mask.shape[0]
    importance = F.conv2d(mask, torch.ones(1, 1, unit, unit), stride=unit)
    k = int(np.floor(mask.shape[2] * mask.shape[3] * patch_budget / unit ** 2))
    flat = importance.view(B, -1)
    vals, idxs = flat.topk(k)
    chosen = torch.zeros_like(flat)
    for row, v, i in zip(chosen, vals, idxs):
        row[i[v > 0]] = 1
    return chosen.view(importance.shape).repeat_interleave(unit, dim=2).repeat_interleave(unit, dim=3)


# ----------------------------------------------------------------------------- next-1
@torch.no_grad()
def collect_failure(model, adv_x, y, universe, targeted, batch_size=128):
    """attack.py:384-406 for one image: ascending list of mask indices the attack fails on."""
    failed = []
    for j0 in range(0, universe.shape[0], batch_size):
        keep = universe[j0:j0 + batch_size]
        preds = model(occlude(adv_x, keep).reshape((-1,) + adv_x.shape[1:])).argmax(-1)
        hit = preds == y.view(-1)[0]
        bad = ~hit if targeted else hit
        failed.extend((bad.nonzero().view(-1) + j0).tolist())
    return failed


# ----------------------------------------------------------------------------- next-2
@torch.no_grad()
def patchcleanser_predict(model, img, single, double, certify=False, batch_size=64):
    """PatchCleanser.robust_predict (PatchCleanser.py:68-97) + robustness_certificate (:102-112)
    for one image (3,H,W); ``single`` (36,1,H,W) / ``double`` (630,1,H,W) bool keep-masks.
    Returns (prediction, certifiable, preds_1, preds_2) like PatchCleanserRecord's fields."""
    def mask(im, msk):                                                      # :99-100
        return im * msk + 0.5 * ~msk

    def certificate(label):                                                 # :102-112
        preds = []
        for i in range(math.ceil(len(double) / batch_size)):
            preds.append(model(mask(img, double[i * batch_size:(i + 1) * batch_size])).argmax(1))
        consistent = torch.cat(preds) == label
        return bool(consistent.all().item()), consistent

    masked = mask(img, single)                                              # :70-71
    preds_1 = model(masked).argmax(1)                                       # :72
    preds_2 = None
    labels, counts = preds_1.unique(sorted=True, return_counts=True)        # :74 (GPU unique sorts)
    label_majority = labels[counts.argmax()].item()                         # :75
    pred = label_majority
    if len(labels) == 1:                                                    # :78-79
        certifiable, preds_2 = certificate(pred)
    else:
        certifiable = False
        for label in labels:                                                # :82-90
            if label == label_majority:
                continue
            for masked_img in masked[preds_1 == label]:
                preds_1_2 = model(mask(masked_img, single)).argmax(-1)
                if (preds_1_2 == label).all():
                    pred = label.item()
    if certify and preds_2 is None:                                         # :93-94
        preds_2 = certificate(label_majority)[1]
    return pred, certifiable, preds_1.numpy(), None if preds_2 is None else preds_2.numpy()
