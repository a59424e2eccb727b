"""Python host wrappers over the C ABI (include/dorpatch_hip.h).

Every function takes/returns ``torch`` CUDA(ROCm) tensors, launches on
``torch.cuda.current_stream()`` and never synchronises.  PyTorch is used for
device memory and streams only; all arithmetic runs in the HIP kernels of
``csrc/dorpatch_hip.hip``.  Non-GPU tensors are rejected: there is no fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import DpNorm, DpUpdateCfg


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a raw handle.  torch's own C entry points where they exist (two C
    calls) instead of building a ``torch.cuda.Stream`` object per launch: at 32 - 64 rows a step is ~240 launches in
    9 - 13 ms and the host's time per launch shows in the step (configs[0]: 9.10 - 9.33 -> 8.74 - 8.94 ms with this and
    the memo in resnetv2._sum_ok, profiles/r06y_host_path_ab.txt)."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _numel(dims):
    n = 1
    for d in dims:
        n *= int(d)
    return n


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"dorpatch_amd.ops: `{name}` must be a GPU tensor "
                           "(no CPU fallback exists; the CPU oracle lives under oracle/ for tests only)")
    if t.dtype != dtype:
        raise TypeError(f"`{name}` must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"`{name}` must be contiguous")
    return t


def require_gpu(t, what):
    """The product path exists only on the GPU: callers that take user tensors (DorPatch.generate,
    PatchCleanser) reject anything else up front, loudly."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("%s needs tensors on a ROCm GPU: the HIP kernels are the only implementation of "
                           "the hot path (no CPU fallback)" % what)
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def make_norm(mean=None, std=None, fill=0.5):
    """dp_norm_t: ``mean``/``std`` None -> raw output with ``fill`` on occluded pixels."""
    n = DpNorm()
    n.fill = float(fill)
    if mean is None:
        n.enable = 0
        for c in range(3):
            n.mean[c], n.std[c] = 0.0, 1.0
    else:
        n.enable = 1
        mean = [float(v) for v in np.asarray(mean, dtype=np.float64).reshape(-1)]
        std = [float(v) for v in np.asarray(std, dtype=np.float64).reshape(-1)]
        if len(mean) == 1:
            mean, std = mean * 3, std * 3
        for c in range(3):
            n.mean[c], n.std[c] = mean[c], std[c]
    return n


RAW_NORM = make_norm(None, None, 0.5)


def debug_set(knob, value):
    """dp_debug_set: launch-geometry overrides for tests / A-B measurements (``_lib.DP_DEBUG_*``); 0 = product default.
    Never called by the product path."""
    _lib.check(_lib.load().dp_debug_set(int(knob), int(value)), "dp_debug_set")


def upload_table(table, device):
    """(n, R, 4) numpy int32 rectangle table -> device int32 tensor."""
    t = np.ascontiguousarray(table, dtype=np.int32)
    assert t.ndim == 3 and t.shape[2] == 4 and 1 <= t.shape[1] <= _lib.DP_MAX_RECTS
    return torch.from_numpy(t).to(device)


# ---------------------------------------------------------------- a-2
def blend(mask, pattern, x, eps, add_x=True, out=None):
    """``utils.clip`` (+ ``+ x``): returns (adv_x | delta, scale (B,), l2 (B,))."""
    lib = _lib.load()
    _chk(mask, torch.float32, "mask"), _chk(pattern, torch.float32, "pattern"), _chk(x, torch.float32, "x")
    B, C, H, W = x.shape
    assert C == 3 and mask.shape == (B, 1, H, W) and pattern.shape == x.shape
    P = H * W
    nchunk = lib.dp_sumsq_nchunk(P)
    partials = torch.empty((B, nchunk), dtype=torch.float32, device=x.device)
    _lib.check(lib.dp_sumsq_partials(_p(mask), _p(pattern), _p(x), B, P, _p(partials), _stream()),
               "dp_sumsq_partials")
    if out is None:
        out = torch.empty_like(x)
    scale = torch.empty((B,), dtype=torch.float32, device=x.device)
    l2 = torch.empty((B,), dtype=torch.float32, device=x.device)
    _lib.check(lib.dp_blend(_p(mask), _p(pattern), _p(x), _p(partials), float(eps), B, P,
                            1 if add_x else 0, _p(out), _p(scale), _p(l2), _stream()), "dp_blend")
    return out, scale, l2


# ---------------------------------------------------------------- a-4 / a-10
def _idx_args(idx, idx2, B):
    _chk(idx, torch.int32, "idx")
    if idx.dim() == 1:
        S, bstride = idx.shape[0], 0
    else:
        assert idx.dim() == 2 and idx.shape[0] == B
        S, bstride = idx.shape[1], idx.shape[1]
    if idx2 is not None:
        _chk(idx2, torch.int32, "idx2")
        assert idx2.shape == idx.shape
    return S, bstride


class KernelTimer(object):
    """A pair of HIP events stamped by the kernel itself (dp_apply_fwd_timed); ``ms()`` blocks."""

    def __init__(self):
        lib = _lib.load()
        self.start, self.stop = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.dp_event_create(ctypes.byref(self.start)), "dp_event_create")
        _lib.check(lib.dp_event_create(ctypes.byref(self.stop)), "dp_event_create")

    def ms(self):
        out = ctypes.c_float()
        _lib.check(_lib.load().dp_event_elapsed_ms(self.start, self.stop, ctypes.byref(out)), "dp_event_elapsed_ms")
        return float(out.value)

    def close(self):
        lib = _lib.load()
        for ev in (self.start, self.stop):
            if ev:
                lib.dp_event_destroy(ev)
        self.start = self.stop = None


def apply_fwd(adv_x, table, idx, idx2=None, norm=RAW_NORM, out=None, timer=None):
    """Occlude (+ normalise) B images under S masks each -> (B*S, 3, H, W).
    ``timer`` (a KernelTimer) makes the launch stamp kernel-begin / kernel-end events."""
    lib = _lib.load()
    _chk(adv_x, torch.float32, "adv_x"), _chk(table, torch.int32, "table")
    B, C, H, W = adv_x.shape
    assert C == 3
    S, bstride = _idx_args(idx, idx2, B)
    if out is None:
        out = torch.empty((B * S, 3, H, W), dtype=torch.float32, device=adv_x.device)
    else:
        _chk(out, torch.float32, "out")
        assert out.numel() == B * S * 3 * H * W
    if timer is None:
        _lib.check(lib.dp_apply_fwd(_p(adv_x), _p(table), table.shape[1], _p(idx), _p(idx2), bstride,
                                    B, S, H, W, ctypes.byref(norm), _p(out), _stream()), "dp_apply_fwd")
    else:
        _lib.check(lib.dp_apply_fwd_timed(_p(adv_x), _p(table), table.shape[1], _p(idx), _p(idx2), bstride,
                                          B, S, H, W, ctypes.byref(norm), _p(out), _stream(), timer.start,
                                          timer.stop), "dp_apply_fwd_timed")
    return out


def apply_bwd(G, table, idx, idx2=None, norm=RAW_NORM, B=None, out=None, accumulate=False):
    """Sum d loss/d out over the S samples of every image -> d loss/d adv_x (B,3,H,W)."""
    lib = _lib.load()
    _chk(G, torch.float32, "G"), _chk(table, torch.int32, "table")
    N, C, H, W = G.shape
    assert C == 3
    if B is None:
        B = idx.shape[0] if idx.dim() == 2 else None
    assert B is not None, "B must be given when idx is shared across images"
    S, bstride = _idx_args(idx, idx2, B)
    assert N == B * S, (N, B, S)
    P = H * W
    nslab = lib.dp_apply_bwd_nslab(B, S, P)
    if out is None:
        assert not accumulate
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=G.device)
    direct = (nslab == 1 and not accumulate)
    slabs = out if direct else torch.empty((nslab, B, 3, H, W), dtype=torch.float32, device=G.device)
    _lib.check(lib.dp_apply_bwd(_p(G), _p(table), table.shape[1], _p(idx), _p(idx2), bstride,
                                B, S, H, W, ctypes.byref(norm), _p(slabs), _stream()), "dp_apply_bwd")
    if not direct:
        _lib.check(lib.dp_sum_slabs(_p(slabs), nslab, B * 3 * P, _p(out), 1 if accumulate else 0,
                                    _stream()), "dp_sum_slabs")
    return out


# ---------------------------------------------------------------- extension: affine placement (dorpatch_amd/placement.py)
def apply_affine_fwd(x, delta, theta, table, idx, idx2=None, norm=RAW_NORM, out=None, timer=None):
    """occlude(norm(x + warp(delta, theta[b,s]))) for B images x S samples -> (B*S,3,H,W); theta (B,S,2,3) fp32.
    ``timer`` (a KernelTimer) makes the launch stamp kernel-begin / kernel-end events."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(delta, torch.float32, "delta"), _chk(theta, torch.float32, "theta")
    _chk(table, torch.int32, "table")
    B, C, H, W = x.shape
    assert C == 3 and delta.shape == x.shape
    S, bstride = _idx_args(idx, idx2, B)
    assert tuple(theta.shape) == (B, S, 2, 3)
    if out is None:
        out = torch.empty((B * S, 3, H, W), dtype=torch.float32, device=x.device)
    if timer is None:
        _lib.check(lib.dp_apply_affine_fwd(_p(x), _p(delta), _p(theta), _p(table), table.shape[1], _p(idx), _p(idx2),
                                           bstride, B, S, H, W, ctypes.byref(norm), _p(out), _stream()),
                   "dp_apply_affine_fwd")
    else:
        _lib.check(lib.dp_apply_affine_fwd_timed(_p(x), _p(delta), _p(theta), _p(table), table.shape[1], _p(idx),
                                                 _p(idx2), bstride, B, S, H, W, ctypes.byref(norm), _p(out), _stream(),
                                                 timer.start, timer.stop), "dp_apply_affine_fwd_timed")
    return out


def apply_affine_bwd(G, theta, theta_inv, table, idx, idx2=None, norm=RAW_NORM, B=None, out=None, accumulate=False):
    """d loss / d delta: the exact adjoint of apply_affine_fwd w.r.t. delta, summed over the S samples."""
    lib = _lib.load()
    _chk(G, torch.float32, "G"), _chk(theta, torch.float32, "theta"), _chk(theta_inv, torch.float32, "theta_inv")
    N, C, H, W = G.shape
    assert C == 3
    if B is None:
        B = idx.shape[0] if idx.dim() == 2 else None
    assert B is not None
    S, bstride = _idx_args(idx, idx2, B)
    assert N == B * S and tuple(theta.shape) == (B, S, 2, 3) and theta_inv.shape == theta.shape
    P = H * W
    nslab = lib.dp_apply_bwd_nslab(B, S, P)
    if out is None:
        assert not accumulate
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=G.device)
    direct = (nslab == 1 and not accumulate)
    slabs = out if direct else torch.empty((nslab, B, 3, H, W), dtype=torch.float32, device=G.device)
    _lib.check(lib.dp_apply_affine_bwd(_p(G), _p(theta), _p(theta_inv), _p(table), table.shape[1], _p(idx), _p(idx2),
                                       bstride, B, S, H, W, ctypes.byref(norm), _p(slabs), _stream()),
               "dp_apply_affine_bwd")
    if not direct:
        _lib.check(lib.dp_sum_slabs(_p(slabs), nslab, B * 3 * P, _p(out), 1 if accumulate else 0, _stream()),
                   "dp_sum_slabs")
    return out


# ---------------------------------------------------------------- a-7
def cw_loss(logits, y, targeted, S, confidence, upstream, loss_out=None, want_grad=True,
            want_pred=True):
    """CW margin loss per row, its gradient w.r.t. the logits, and the argmax."""
    lib = _lib.load()
    _chk(logits, torch.float32, "logits"), _chk(y, torch.int64, "y"), _chk(targeted, torch.int32, "targeted")
    N, C = logits.shape
    assert N % S == 0 and y.numel() * S >= N
    loss = loss_out if loss_out is not None else torch.empty((N,), dtype=torch.float32, device=logits.device)
    _chk(loss, torch.float32, "loss_out")
    dlogits = torch.empty_like(logits) if want_grad else None
    pred = torch.empty((N,), dtype=torch.int32, device=logits.device) if want_pred else None
    _lib.check(lib.dp_cw_loss(_p(logits), _p(y), _p(targeted), N, C, S, float(confidence),
                              float(upstream), _p(loss), _p(dlogits), _p(pred), _stream()), "dp_cw_loss")
    return loss, dlogits, pred


def argmax(logits):
    lib = _lib.load()
    _chk(logits, torch.float32, "logits")
    N, C = logits.shape
    pred = torch.empty((N,), dtype=torch.int32, device=logits.device)
    _lib.check(lib.dp_argmax(_p(logits), N, C, _p(pred), _stream()), "dp_argmax")
    return pred


# ---------------------------------------------------------------- a-5
def local_variance(x):
    lib = _lib.load()
    _chk(x, torch.float32, "x")
    B, C, H, W = x.shape
    assert C == 3
    lv = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    _lib.check(lib.dp_local_variance(_p(x), B, H, W, _p(lv), _stream()), "dp_local_variance")
    return lv


def struct_loss(adv_x, lv_x, out=None):
    """loss_struc (B,) = mean_{h,w}( mean_c(L(adv_x)) / (lv_x + 1e-5) )."""
    lib = _lib.load()
    _chk(adv_x, torch.float32, "adv_x"), _chk(lv_x, torch.float32, "lv_x")
    B, C, H, W = adv_x.shape
    ntile = lib.dp_struct_ntile(H, W)
    partials = torch.empty((B, ntile), dtype=torch.float32, device=adv_x.device)
    _lib.check(lib.dp_struct_loss(_p(adv_x), _p(lv_x), B, H, W, _p(partials), _stream()), "dp_struct_loss")
    if out is None:
        out = torch.empty((B,), dtype=torch.float32, device=adv_x.device)
    _lib.check(lib.dp_reduce_rows(_p(partials), B, ntile, 1.0 / float(H * W), _p(out), _stream()),
               "dp_reduce_rows")
    return out


# ---------------------------------------------------------------- a-6
def mask_grid(H, W, unit, win):
    return ((H - unit) // unit + 1, (W - unit) // unit + 1, (H - win) // win + 1, (W - win) // win + 1)


def mask_stats(mask, unit, win, gl_out=None, dens_out=None):
    """cell_sumsq (B,ncy,ncx), win_sum (B,nwy,nwx), group_lasso (B,), density (B,)."""
    lib = _lib.load()
    _chk(mask, torch.float32, "mask")
    B, C, H, W = mask.shape
    assert C == 1
    ncy, ncx, nwy, nwx = mask_grid(H, W, unit, win)
    dev = mask.device
    cell = torch.empty((B, ncy, ncx), dtype=torch.float32, device=dev)
    wsum = torch.empty((B, nwy, nwx), dtype=torch.float32, device=dev)
    gl = gl_out if gl_out is not None else torch.empty((B,), dtype=torch.float32, device=dev)
    dens = dens_out if dens_out is not None else torch.empty((B,), dtype=torch.float32, device=dev)
    _lib.check(lib.dp_mask_stats(_p(mask), B, H, W, unit, win, _p(cell), _p(wsum), _p(gl), _p(dens),
                                 _stream()), "dp_mask_stats")
    return cell, wsum, gl, dens


# ---------------------------------------------------------------- a-2 bwd, a-5/a-6 grads, a-9
def project_update(x, adv_x, lv_x, g_adv, scale, structured, pattern, mask, *, stage, lr=None,
                   coeff_gl=None, cell_sumsq=None, win_sum=None, unit=7, win=28, density=0.0,
                   clip_min=0.0, clip_max=1.0, save_best=None, best_pattern=None, best_mask=None,
                   do_update=True, want_grads=False):
    """Fused chain rule + regulariser gradients + signed update (in place on pattern/mask)."""
    lib = _lib.load()
    for name, t in (("x", x), ("adv_x", adv_x), ("lv_x", lv_x), ("g_adv", g_adv), ("scale", scale),
                    ("structured", structured), ("pattern", pattern), ("mask", mask)):
        _chk(t, torch.float32, name)
    B, C, H, W = x.shape
    cfg = DpUpdateCfg(B=B, H=H, W=W, stage=int(stage), unit=int(unit), win=int(win),
                      do_update=1 if do_update else 0, density=float(density),
                      clip_min=float(clip_min), clip_max=float(clip_max))
    for name, t in (("lr", lr), ("coeff_gl", coeff_gl), ("cell_sumsq", cell_sumsq),
                    ("win_sum", win_sum), ("best_pattern", best_pattern), ("best_mask", best_mask)):
        if t is not None:
            _chk(t, torch.float32, name)
    if save_best is not None:
        _chk(save_best, torch.int32, "save_best")
    gp = torch.empty_like(pattern) if want_grads else None
    gm = torch.empty_like(mask) if want_grads else None
    _lib.check(lib.dp_project_update(ctypes.byref(cfg), _p(x), _p(adv_x), _p(lv_x), _p(g_adv),
                                     _p(scale), _p(structured), _p(coeff_gl), _p(lr),
                                     _p(cell_sumsq), _p(win_sum), _p(save_best), _p(pattern),
                                     _p(mask), _p(best_pattern), _p(best_mask), _p(gp), _p(gm),
                                     _stream()), "dp_project_update")
    return gp, gm


# ---------------------------------------------------------------- a-8: 3x3 convolutions on the matrix cores
CONV3X3_SIDES = (56, 28, 14, 7, 96, 48, 24, 12)      # planes of ResNetV2-50 at 224 x 224 and at 384 x 384

# bench.py's "roofline_conv": set to a list for ONE extra, untimed step and every matrix-core convolution launch appends
# (kernel, shape key, flop, start event, stop event) — torch events on the launch stream (the kernels run on torch's current
# stream).  None (always, in the product): no events, no overhead.
CONV_EVENTS = None


def _timed_conv(kernel, key, flop, launch):
    if CONV_EVENTS is None:
        return launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = launch()
    b.record()
    CONV_EVENTS.append((kernel, key, flop, a, b))
    return out


def conv3x3_supported(x, weight, stride=(1, 1), padding=(1, 1)):
    """Shapes dp_conv3x3_fwd takes: fp32 GPU NCHW, 3x3 / stride 1 / pad 1, square planes of side 56 / 28 / 14 / 7 or
    96 / 48 / 24 / 12, C % 8 == 0, O % 64 == 0 (every stride-1 3x3 convolution of ResNetV2-50 at 224 x 224 and 384 x 384)."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and x.shape[2] == x.shape[3] and x.shape[2] in CONV3X3_SIDES and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 8 == 0 and weight.shape[0] % 64 == 0
            and x.shape[0] * max(weight.shape[0], weight.shape[1]) * x.shape[2] * x.shape[3] < 2 ** 31)


def pack_conv3x3_weights(w, transpose=False):
    """(O, C, 3, 3) frozen weights -> the k-walk order of dp_conv3x3_fwd: [og][chunk][cp][kh][kw][half][o] with output
    channel 64 og + o and input channel 8 chunk + 2 cp + half (include/dorpatch_hip.h).  ``transpose``: the weights of the
    INPUT-GRADIENT convolution instead (w'[c][o][kh][kw] = w[o][c][2-kh][2-kw]: dx = conv3x3(dy, w')).  Plain tensor
    reshuffle, done once per frozen convolution."""
    w = w.detach().float()
    if transpose:
        w = w.flip(2, 3).transpose(0, 1)
    O, C = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 3) and C % 8 == 0 and O % 64 == 0
    return w.reshape(O // 64, 64, C // 8, 4, 2, 3, 3).permute(0, 2, 3, 5, 6, 4, 1).contiguous()


CONV3X3_FOLD_SIDES = (56, 28, 14, 96, 48, 24, 12)     # every side but 7 (rows of 7 floats are not 16-byte multiples)


def conv3x3_fwd(x, wt, ab=None):
    """y = conv2d(x', w, stride 1, padding 1) for x (N,C,S,S), wt = pack_conv3x3_weights(w), on v_mfma_f32_32x32x2_f32.
    ``ab`` (N,C,2) from ``gn_stats``: x' = relu(group_norm(x)) applied while staging (x is the RAW tensor; every side but 7)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(wt, torch.float32, "wt")
    N, C, H, W = x.shape
    O = wt.shape[0] * wt.shape[-1]
    assert wt.numel() == C * 9 * O
    y = torch.empty((N, O, H, W), dtype=torch.float32, device=x.device)
    if ab is not None:
        _chk(ab, torch.float32, "ab")
        assert ab.numel() == N * C * 2

    def launch():
        if ab is None:
            _lib.check(lib.dp_conv3x3_fwd(_p(x), _p(wt), N, C, O, H, W, _p(y), _stream()), "dp_conv3x3_fwd")
        else:
            _lib.check(lib.dp_conv3x3_gn_fwd(_p(x), _p(wt), _p(ab), N, C, O, H, W, _p(y), _stream()), "dp_conv3x3_gn_fwd")
        return y
    return _timed_conv("k_conv3x3_mfma", (N, C, O, H * W, ab is not None, False), 18.0 * N * H * W * C * O, launch)


# ---------------------------------------------------------------- a-8 (round 6): Winograd F(2x2, 3x3) on the matrix cores
CONV3X3_WINO_SIDES = (56, 28, 14, 7, 96, 48, 24, 12)
WINO_CH = 4            # input channels per K-chunk of k_conv3x3_wino (kWnCh in csrc/conv3x3_wino.inc): fixes the packed layout


def conv3x3_wino_supported(x, weight, stride=(1, 1), padding=(1, 1)):
    """Shapes dp_conv3x3_wino_fwd takes: fp32 GPU NCHW, 3x3 / stride 1 / pad 1, square planes of side 56 / 28 / 14 / 7 (224 x 224 inputs) or
    96 / 48 / 24 / 12 (384 x 384), C % 8 == 0, O % 64 == 0."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and x.shape[2] == x.shape[3] and x.shape[2] in CONV3X3_WINO_SIDES and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 8 == 0 and weight.shape[0] % 64 == 0
            and x.shape[0] * max(weight.shape[0], weight.shape[1]) * x.shape[2] * x.shape[3] < 2 ** 31)


def pack_conv3x3_wino_weights(w, transpose=False):
    """(O, C, 3, 3) frozen weights -> U = G g G^T, the 4 x 4 Winograd-domain filter of F(2 x 2, 3 x 3), in
    dp_conv3x3_wino_fwd's OPERAND order [og][chunk][position 4 xi + nu][half][lane][k-step t][fragment f] (output channel
    64 og + 32 f + lane, input channel 4 chunk + 2 t + half: the float4 a lane loads is its four A operands of a
    (chunk, position); include/dorpatch_hip.h).  Computed in fp64 and rounded once.  ``transpose``: the filter of the INPUT
    GRADIENT (w'[c][o][kh][kw] = w[o][c][2 - kh][2 - kw]).  Once per frozen convolution."""
    w = w.detach().double()
    if transpose:
        w = w.transpose(0, 1).flip(2, 3)
    O, C = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 3) and C % 8 == 0 and O % 64 == 0
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    # (og, f, l, chunk, t, half, p): output channel 64 og + 32 f + l, input channel WINO_CH chunk + 2 t + half
    U = torch.einsum("ai,ocij,bj->ocab", G, w, G).reshape(O // 64, 2, 32, C // WINO_CH, WINO_CH // 2, 2, 16)
    return U.permute(0, 3, 6, 5, 2, 4, 1).contiguous().float()      # [og][chunk][p][half][l][t][f]


def conv3x3_wino_fwd(x, wt, ab=None):
    """y = conv2d(x', w, stride 1, padding 1) for x (N,C,S,S), wt = pack_conv3x3_wino_weights(w): Winograd F(2x2, 3x3) with
    the 16 position GEMMs on v_mfma_f32_32x32x2_f32 (2.25x fewer multiplications than the direct form; fp32, fixed order,
    ~1e-6 of the output scale away from the direct kernels).  ``ab`` (N,C,2) from ``gn_stats``: x' = relu(group_norm(x))
    applied while staging (x is the RAW tensor; the zero padding pads the normalised activation)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(wt, torch.float32, "wt")
    N, C, H, W = x.shape
    O = wt.shape[0] * 64
    assert wt.numel() == C * 16 * O, (wt.shape, C, O)
    y = torch.empty((N, O, H, W), dtype=torch.float32, device=x.device)
    if ab is not None:
        _chk(ab, torch.float32, "ab")
        assert ab.numel() == N * C * 2

    def launch():
        _lib.check(lib.dp_conv3x3_wino_fwd(_p(x), _p(wt), _p(ab), N, C, O, H, W, _p(y), _stream()), "dp_conv3x3_wino_fwd")
        return y
    return _timed_conv("k_conv3x3_wino", (N, C, O, H * W, ab is not None, False), 18.0 * N * H * W * C * O, launch)


CONV3X3S2_SIDES = (56, 28, 14)        # INPUT sides of the stride-2 kernel (ResNetV2-50 at 224 x 224)


def conv3x3s2_supported(x, weight, stride=(2, 2), padding=(1, 1)):
    """Shapes dp_conv3x3s2_fwd takes: fp32 GPU NCHW, 3x3 / stride 2 / pad 1, square planes of side 56 / 28 / 14,
    C % 8 == 0, O % 64 == 0 (conv2 of the first bottleneck of stages 2-4 of ResNetV2-50 at 224 x 224)."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and x.shape[2] == x.shape[3] and x.shape[2] in CONV3X3S2_SIDES and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 8 == 0 and weight.shape[0] % 64 == 0
            and x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] < 2 ** 31)


def conv3x3s2_fwd(x, wt, ab=None):
    """y = conv2d(x', w, stride 2, padding 1) for x (N,C,S,S), S in 56 / 28 / 14, wt = pack_conv3x3_weights(w), on
    v_mfma_f32_32x32x2_f32 (same summation order as conv3x3_fwd).  ``ab`` (N,C,2) from ``gn_stats``:
    x' = relu(group_norm(x)) applied while staging (x is the RAW tensor)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(wt, torch.float32, "wt")
    N, C, H, W = x.shape
    O = wt.shape[0] * wt.shape[-1]
    assert wt.numel() == C * 9 * O
    y = torch.empty((N, O, H // 2, W // 2), dtype=torch.float32, device=x.device)
    if ab is not None:
        _chk(ab, torch.float32, "ab")
        assert ab.numel() == N * C * 2

    def launch():
        _lib.check(lib.dp_conv3x3s2_fwd(_p(x), _p(wt), _p(ab), N, C, O, H, W, _p(y), _stream()), "dp_conv3x3s2_fwd")
        return y
    return _timed_conv("k_conv3x3s2_mfma", (N, C, O, H * W // 4, ab is not None, False), 18.0 * N * (H // 2) * (W // 2) * C * O,
                       launch)


CONV3X3S2_BWD_SIDES = (28, 14, 7, 48, 24, 12)      # sides of dy (the OUTPUT of the strided convolution)


def conv3x3s2_bwd_supported(dy, weight, stride=(2, 2), padding=(1, 1)):
    """Shapes dp_conv3x3s2_bwd takes: dy (N,O,Ho,Ho) fp32 GPU NCHW of a 3x3 / stride 2 / pad 1 convolution with an EVEN
    input side 2 Ho, Ho in 28 / 14 / 7 (224 x 224 inputs) or 48 / 24 / 12 (384 x 384), O % 16 == 0, C % 64 == 0."""
    return (isinstance(dy, torch.Tensor) and dy.is_cuda and dy.dtype == torch.float32 and dy.dim() == 4 and dy.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and dy.shape[2] == dy.shape[3] and dy.shape[2] in CONV3X3S2_BWD_SIDES and weight.shape[0] == dy.shape[1]
            and weight.shape[0] % 16 == 0 and weight.shape[1] % 64 == 0
            and dy.shape[0] * max(weight.shape[0], 4 * weight.shape[1]) * dy.shape[2] * dy.shape[3] < 2 ** 31)


def pack_conv3x3s2_dgrad_weights(w, pairs=True):
    """(O, C, 3, 3) frozen weights of a 3x3 / stride 2 / pad 1 convolution -> the parity classes of its INPUT GRADIENT in
    dp_conv3x3s2_bwd's k-walk order (include/dorpatch_hip.h).  Class (pr, pc) of dx — rows 2a + pr, columns 2b + pc — sums
    dy[a + th][b + tw] * w[.][.][kh][kw] over th <= pr, tw <= pc with kh = 1 (pr = 0) or (2, 0)[th], kw likewise.
    ``pairs`` (DP_S2BWD_PAIRS, the default form): both column classes of a row parity share a workgroup — row class 1 then 0,
    per (chunk of 8 dy channels, channel pair, th) the three weight vectors kw = (1, 2, 0); else (DP_S2BWD_CLASSES) the four
    classes (1,1), (0,1), (1,0), (0,0) back to back.  Plain tensor reshuffle, once per frozen convolution."""
    w = w.detach().float()
    O, C = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 3) and O % 16 == 0 and C % 64 == 0
    parts = []
    if pairs:
        for pr in (1, 0):
            khs = [2, 0] if pr else [1]
            ws = w[:, :, khs][:, :, :, [1, 2, 0]]                                  # (O, C, th, j)
            ws = ws.reshape(O // 8, 4, 2, C // 64, 64, len(khs), 3)                 # chunk, cp, half, og, c', th, j
            parts.append(ws.permute(3, 0, 1, 5, 6, 2, 4).contiguous().reshape(-1))
        return torch.cat(parts)
    for pr, pc in ((1, 1), (0, 1), (1, 0), (0, 0)):
        khs, kws = ([2, 0] if pr else [1]), ([2, 0] if pc else [1])
        T = len(khs) * len(kws)
        CH = 16 // T
        ws = w[:, :, khs][:, :, :, kws]                                   # (O, C, th, tw)
        ws = ws.reshape(O // CH, CH // 2, 2, C // 64, 64, len(khs), len(kws))      # chunk, cp, half, og, c', th, tw
        parts.append(ws.permute(3, 0, 1, 5, 6, 2, 4).contiguous().reshape(-1))
    return torch.cat(parts)


def conv3x3s2_bwd(dy, wt, C, pairs=True):
    """dx (N,C,2Ho,2Ho) = the input gradient of conv2d(x, w, stride 2, padding 1) for dy (N,O,Ho,Ho) and
    wt = pack_conv3x3s2_dgrad_weights(w, pairs), on v_mfma_f32_32x32x2_f32 (exact f32, fixed order: deterministic; both
    forms give the same bits)."""
    lib = _lib.load()
    _chk(dy, torch.float32, "dy"), _chk(wt, torch.float32, "wt")
    N, O, Ho, Wo = dy.shape
    assert wt.numel() == 9 * C * O, (wt.shape, C, O)
    dx = torch.empty((N, C, 2 * Ho, 2 * Wo), dtype=torch.float32, device=dy.device)
    form = _lib.DP_S2BWD_PAIRS if pairs else _lib.DP_S2BWD_CLASSES

    def launch():
        _lib.check(lib.dp_conv3x3s2_bwd(_p(dy), _p(wt), N, O, C, Ho, Wo, _p(dx), form, _stream()), "dp_conv3x3s2_bwd")
        return dx
    return _timed_conv("k_conv3x3s2_dgrad2" if pairs else "k_conv3x3s2_dgrad", (N, O, C, Ho * Wo, False, False),
                       18.0 * N * Ho * Wo * C * O, launch)


# ---------------------------------------------------------------- a-8: 1x1 convolutions on the matrix cores (round 5)
def conv1x1_supported(x, weight):
    """Shapes dp_conv1x1_fwd takes: fp32 GPU NCHW, (O, C, 1, 1) filter with C % 16 == 0 and O % 64 == 0, planes of
    H*W % 4 == 0 pixels or 7 x 7 (every stride-1 1x1 convolution of ResNetV2-50 at 224 x 224 and 384 x 384)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    HW = x.shape[2] * x.shape[3]
    return (weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1) and weight.shape[1] == x.shape[1]
            and weight.shape[1] % 16 == 0 and weight.shape[0] % 64 == 0 and (HW % 4 == 0 or HW == 49)
            and x.shape[0] * max(weight.shape[0], weight.shape[1]) * HW < 2 ** 31)


def pack_conv1x1_weights(w, transpose=False):
    """(O, C[, 1, 1]) frozen weights -> the k-walk order of dp_conv1x1_fwd: [og][chunk][k][o] with output channel 64 og + o
    and input channel 16 chunk + k (include/dorpatch_hip.h).  ``transpose``: the weights of the INPUT-GRADIENT convolution
    (dx = conv1x1(dy, w^T)).  Plain tensor reshuffle, done once per frozen convolution."""
    w = w.detach().float().reshape(w.shape[0], w.shape[1])
    if transpose:
        w = w.t()
    O, C = w.shape
    assert C % 16 == 0 and O % 64 == 0
    return w.reshape(O // 64, 64, C // 16, 16).permute(0, 2, 3, 1).contiguous()


def conv1x1_fwd(x, wt, ab=None, res=None, out=None):
    """y = conv2d(x', w) [+ res] for x (N,C,H,W) and wt = pack_conv1x1_weights(w), on v_mfma_f32_32x32x2_f32.
    ``ab`` (N,C,2) from ``gn_stats``: x' = relu(group_norm(x)) applied while staging (x is the RAW tensor), else x' = x.
    ``res`` (N,O,H,W) is added in the epilogue; ``out`` may be ``res`` itself (in-place accumulation)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(wt, torch.float32, "wt")
    N, C, H, W = x.shape
    O = wt.shape[0] * wt.shape[-1]
    assert wt.numel() == C * O, (wt.shape, C, O)
    if ab is not None:
        _chk(ab, torch.float32, "ab")
        assert ab.numel() == N * C * 2
    if res is not None:
        _chk(res, torch.float32, "res")
        assert tuple(res.shape) == (N, O, H, W)
    if out is None:
        out = torch.empty((N, O, H, W), dtype=torch.float32, device=x.device)
    else:
        _chk(out, torch.float32, "out")
        assert tuple(out.shape) == (N, O, H, W)
    def launch():
        _lib.check(lib.dp_conv1x1_fwd(_p(x), _p(wt), _p(ab), _p(res), N, C, O, H * W, _p(out), _stream()), "dp_conv1x1_fwd")
        return out
    return _timed_conv("k_conv1x1_mfma", (N, C, O, H * W, ab is not None, res is not None), 2.0 * N * H * W * C * O, launch)


def gn_stats(x, weight, bias, groups, eps, res=None, stats_out=None):
    """Statistics-only GroupNorm pass: (mean, rstd, ab, s) with s = x (+ res) and ab (N,C,2) the affine coefficients of
    ``relu(group_norm(s))`` for a consumer that applies them itself (``conv1x1_fwd(..., ab=ab)``).
    ``stats_out`` = (mean, rstd) views of N*groups floats to write the statistics into."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(weight, torch.float32, "weight"), _chk(bias, torch.float32, "bias")
    N, C = x.shape[0], x.shape[1]
    HW = _numel(x.shape[2:])
    if stats_out is None:
        mean = torch.empty((N * groups,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    else:
        mean, rstd = stats_out
        _chk(mean, torch.float32, "mean"), _chk(rstd, torch.float32, "rstd")
        assert mean.numel() == N * groups and rstd.numel() == N * groups
    ab = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
    ssum = None
    if res is not None:
        _chk(res, torch.float32, "res")
        assert res.shape == x.shape
        ssum = torch.empty_like(x)
    _lib.check(lib.dp_gn_stats(_p(x), _p(res), _p(ssum), _p(weight), _p(bias), N, C, HW, int(groups), float(eps),
                               _p(mean), _p(rstd), _p(ab), _stream()), "dp_gn_stats")
    return mean, rstd, ab, (x if res is None else ssum)


# ---------------------------------------------------------------- a-8: fused GroupNorm + ReLU (frozen backbone)
def gn_relu_supported(x, groups):
    """Shapes the fused kernel accepts: fp32 GPU NCHW, (C/groups)*H*W a multiple of 4 and < 2^20."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2):
        return False
    C = x.shape[1]
    if C % groups:
        return False
    L = (C // groups) * _numel(x.shape[2:])
    return L % 4 == 0 and L < (1 << 20)


def gn_relu_fwd(x, weight, bias, groups, eps, res=None, stats_out=None):
    """s = x (+ res);  y = relu(group_norm(s)).  Returns (y, mean, rstd, s); mean / rstd are the
    per-(sample, group) statistics the backward needs; s is x itself when res is None.
    ``stats_out`` = (mean, rstd) views of N*groups floats to write the statistics into."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(weight, torch.float32, "weight"), _chk(bias, torch.float32, "bias")
    N, C = x.shape[0], x.shape[1]
    HW = _numel(x.shape[2:])
    y = torch.empty_like(x)
    if stats_out is None:
        mean = torch.empty((N * groups,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    else:
        mean, rstd = stats_out
        _chk(mean, torch.float32, "mean"), _chk(rstd, torch.float32, "rstd")
        assert mean.numel() == N * groups and rstd.numel() == N * groups
    ssum = None
    if res is not None:
        _chk(res, torch.float32, "res")
        assert res.shape == x.shape
        ssum = torch.empty_like(x)
    _lib.check(lib.dp_gn_relu_fwd(_p(x), _p(res), _p(ssum), _p(weight), _p(bias), N, C, HW, int(groups),
                                  float(eps), _p(y), _p(mean), _p(rstd), _stream()), "dp_gn_relu_fwd")
    return y, mean, rstd, (x if res is None else ssum)


def gn_relu_bwd(dy, x, weight, bias, mean, rstd, groups, dres=None):
    """d loss / d s of y = relu(group_norm(s)) (+ dres, the gradient reaching s through the shortcut).
    Weights frozen: no gamma / beta gradients."""
    lib = _lib.load()
    _chk(dy, torch.float32, "dy"), _chk(x, torch.float32, "x")
    if dres is not None:
        _chk(dres, torch.float32, "dres")
    N, C = x.shape[0], x.shape[1]
    HW = _numel(x.shape[2:])
    dx = torch.empty_like(x)
    _lib.check(lib.dp_gn_relu_bwd(_p(dy), _p(dres), _p(x), _p(weight), _p(bias), _p(mean), _p(rstd), N, C, HW,
                                  int(groups), _p(dx), _stream()), "dp_gn_relu_bwd")
    return dx


def gn_relu_bwd_gather(dy, x_tabs, tab_rows, smap, weight, bias, mean, rstd, groups, dres=None):
    """``gn_relu_bwd`` over M selected samples: output sample n takes x / mean / rstd of SOURCE sample ``smap[n]``.
    ``x_tabs``: the GroupNorm inputs saved by the micro-batches of one forward, a list of (rows, C, ...) tensors, every
    one but the last with ``tab_rows`` rows (source sample s = row s % tab_rows of x_tabs[s // tab_rows]);
    ``mean`` / ``rstd``: (total source samples * groups,), indexed by source sample.  dy / dres / result: (M, C, ...)."""
    lib = _lib.load()
    _chk(dy, torch.float32, "dy"), _chk(smap, torch.int32, "smap")
    _chk(mean, torch.float32, "mean"), _chk(rstd, torch.float32, "rstd")
    if dres is not None:
        _chk(dres, torch.float32, "dres")
    for t in x_tabs:
        _chk(t, torch.float32, "x_tabs[k]")
        assert t.shape[1:] == dy.shape[1:] and t.shape[0] <= tab_rows
    assert all(t.shape[0] == tab_rows for t in x_tabs[:-1]) and smap.numel() == dy.shape[0]
    M, C = dy.shape[0], dy.shape[1]
    HW = _numel(dy.shape[2:])
    ptrs = (ctypes.c_void_p * len(x_tabs))(*[t.data_ptr() for t in x_tabs])
    dx = torch.empty_like(dy)
    _lib.check(lib.dp_gn_relu_bwd_gather(_p(dy), _p(dres), ptrs, len(x_tabs), int(tab_rows), _p(smap), _p(weight),
                                         _p(bias), _p(mean), _p(rstd), M, C, HW, int(groups), _p(dx), _stream()),
               "dp_gn_relu_bwd_gather")
    return dx


class GnReluFunction(torch.autograd.Function):
    """autograd node over dp_gn_relu_fwd / dp_gn_relu_bwd.  Saves only the GN input and 2*N*G
    statistics (eager PyTorch also keeps the ReLU output)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        x = x.contiguous()
        y, mean, rstd, _ = gn_relu_fwd(x, weight, bias, groups, eps)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        dx = gn_relu_bwd(dy.contiguous(), x, weight, bias, mean, rstd, ctx.groups)
        return dx, None, None, None, None


class AddGnReluFunction(torch.autograd.Function):
    """(x, res) -> (s, y) with s = x + res and y = relu(group_norm(s)): the bottleneck's residual add
    fused into the next GroupNorm+ReLU (one kernel each way).  In the backward the gradient arriving
    at s through the shortcut is added inside dp_gn_relu_bwd, replacing autograd's accumulation add."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, groups, eps):
        y, mean, rstd, s = gn_relu_fwd(x.contiguous(), weight, bias, groups, eps, res=res.contiguous())
        ctx.save_for_backward(s, weight, bias, mean, rstd)
        ctx.groups = groups
        ctx.set_materialize_grads(False)
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, weight, bias, mean, rstd = ctx.saved_tensors
        if dy is None:
            dx = ds
        else:
            dx = gn_relu_bwd(dy.contiguous(), s, weight, bias, mean, rstd, ctx.groups,
                             dres=None if ds is None else ds.contiguous())
        return dx, dx, None, None, None, None


# ---------------------------------------------------------------- a-8: fused zero-pad(1) + maxpool 3x3/2 (BiT stem)
def pad_maxpool_supported(x):
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0)


def pad_maxpool_fwd(x):
    """max_pool2d(pad(x, 1, value=0), 3, stride=2) -> (y, code uint8 argmax codes for the backward)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    code = torch.empty((N, C, H // 2, W // 2), dtype=torch.uint8, device=x.device)
    _lib.check(lib.dp_pad_maxpool_fwd(_p(x), N * C, H, W, _p(y), _p(code), _stream()), "dp_pad_maxpool_fwd")
    return y, code


def pad_maxpool_bwd(dy, code, H, W):
    lib = _lib.load()
    _chk(dy, torch.float32, "dy"), _chk(code, torch.uint8, "code")
    if dy.data_ptr() % 16:          # a dense view at an odd storage offset: the kernel reads dy in 16-byte words
        dy = dy.clone()
    N, C = dy.shape[0], dy.shape[1]
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    _lib.check(lib.dp_pad_maxpool_bwd(_p(dy), _p(code), N * C, H, W, _p(dx), _stream()), "dp_pad_maxpool_bwd")
    return dx


class PadMaxPoolFunction(torch.autograd.Function):
    """autograd node over dp_pad_maxpool_fwd / _bwd; saves 1 byte per output element."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y, code = pad_maxpool_fwd(x)
        ctx.save_for_backward(code)
        ctx.hw = (x.shape[2], x.shape[3])
        return y

    @staticmethod
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        return pad_maxpool_bwd(dy.contiguous(), code, *ctx.hw)


# ---------------------------------------------------------------- a-8: stem conv (7x7/2, 3 input channels) input gradient
def stem_dgrad_supported(x, weight, stride, padding):
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and tuple(weight.shape[1:]) == (3, 7, 7) and tuple(stride) == (2, 2) and tuple(padding) == (3, 3)
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[0] <= 65535)


def stem_dgrad(dy, weight):
    """d loss / d input of conv2d(input, weight (K,3,7,7), stride 2, padding 3) given dy (N,K,Ho,Wo)."""
    lib = _lib.load()
    _chk(dy, torch.float32, "dy"), _chk(weight, torch.float32, "weight")
    N, K, Ho, Wo = dy.shape
    assert tuple(weight.shape) == (K, 3, 7, 7)
    dx = torch.empty((N, 3, 2 * Ho, 2 * Wo), dtype=torch.float32, device=dy.device)
    _lib.check(lib.dp_stem_dgrad(_p(dy), _p(weight), N, K, Ho, Wo, _p(dx), _stream()), "dp_stem_dgrad")
    return dx


def stem_dgrad_reduce(dy, weight, table, idx, idx2=None, norm=RAW_NORM, B=None, out=None, accumulate=False):
    """apply_bwd(stem_dgrad(dy, weight), ...) in one launch: the per-sample (B*S,3,H,W) input gradient is never
    materialised.  Same arguments / result as ``apply_bwd`` with ``G`` replaced by the stem-conv output gradient."""
    lib = _lib.load()
    _chk(dy, torch.float32, "dy"), _chk(weight, torch.float32, "weight"), _chk(table, torch.int32, "table")
    N, K, Ho, Wo = dy.shape
    assert tuple(weight.shape) == (K, 3, 7, 7)
    if B is None:
        B = idx.shape[0] if idx.dim() == 2 else None
    assert B is not None, "B must be given when idx is shared across images"
    S, bstride = _idx_args(idx, idx2, B)
    assert N == B * S, (N, B, S)
    H, W = 2 * Ho, 2 * Wo
    nslab = lib.dp_apply_bwd_nslab(B, S, H * W)
    if out is None:
        assert not accumulate
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=dy.device)
    direct = (nslab == 1 and not accumulate)
    slabs = out if direct else torch.empty((nslab, B, 3, H, W), dtype=torch.float32, device=dy.device)
    _lib.check(lib.dp_stem_dgrad_reduce(_p(dy), _p(weight), _p(table), table.shape[1], _p(idx), _p(idx2), bstride,
                                        B, S, K, Ho, Wo, ctypes.byref(norm), _p(slabs), _stream()),
               "dp_stem_dgrad_reduce")
    if not direct:
        _lib.check(lib.dp_sum_slabs(_p(slabs), nslab, B * 3 * H * W, _p(out), 1 if accumulate else 0, _stream()),
                   "dp_sum_slabs")
    return out


def stem_conv_supported(x, weight, stride=(2, 2), padding=(3, 3)):
    """Shapes dp_stem_conv_fwd takes: fp32 GPU NCHW (N,3,H,224), H even, filter (64,3,7,7) / stride 2 / pad 3."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and tuple(weight.shape) == (64, 3, 7, 7) and tuple(stride) == (2, 2) and tuple(padding) == (3, 3)
            and x.shape[1] == 3 and x.shape[3] == 224 and x.shape[2] % 2 == 0 and x.numel() < 2 ** 31)


def pack_stem_weights(w):
    """(64, 3, 7, 7) frozen stem filter -> dp_stem_conv_fwd's k-walk: [7 j + kw][half][o] = w[o][r % 3][r // 3][kw] with
    r = 2 j + half over the 21 (kh, channel) rows, the 22nd is zero padding (include/dorpatch_hip.h)."""
    w = w.detach().float()
    assert tuple(w.shape) == (64, 3, 7, 7)
    rows = w.permute(2, 1, 3, 0).reshape(21, 7, 64)                     # [3 kh + c][kw][o]
    rows = torch.cat([rows, torch.zeros_like(rows[:1])])                 # r = 21: padding
    return rows.reshape(11, 2, 7, 64).permute(0, 2, 1, 3).contiguous()   # [j][kw][half][o]


def stem_conv_fwd(x, wt):
    """y (N,64,H/2,112) = conv2d(x, w, stride 2, padding 3) for x (N,3,H,224), wt = pack_stem_weights(w), on
    v_mfma_f32_32x32x2_f32."""
    lib = _lib.load()
    _chk(x, torch.float32, "x"), _chk(wt, torch.float32, "wt")
    N, C, H, W = x.shape
    assert C == 3 and wt.numel() == 77 * 2 * 64
    y = torch.empty((N, 64, H // 2, W // 2), dtype=torch.float32, device=x.device)

    def launch():
        _lib.check(lib.dp_stem_conv_fwd(_p(x), _p(wt), N, H, W, _p(y), _stream()), "dp_stem_conv_fwd")
        return y
    return _timed_conv("k_stem_conv_mfma", (N, 3, 64, (H // 2) * (W // 2), False, False), 2.0 * N * (H // 2) * (W // 2) * 147 * 64,
                       launch)


class StemConvFunction(torch.autograd.Function):
    """Stem convolution with a frozen filter: forward through MIOpen, input gradient through
    dp_stem_dgrad (the 3-channel transposed convolution libraries handle poorly)."""

    @staticmethod
    def forward(ctx, x, weight):
        from . import libconv
        ctx.save_for_backward(weight)
        return libconv.conv_fwd(x, weight, (2, 2), (3, 3))

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        return stem_dgrad(dy.contiguous(), weight.contiguous()), None


# ---------------------------------------------------------------- a-8: strided 1x1 (downsample) convolutions
def subsample2_supported(x):
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.is_contiguous())


def subsample2(x):
    """x[:, :, ::2, ::2] as a dense tensor (even pixels only: what a stride-2 1x1 convolution reads)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(lib.dp_subsample2(_p(x), N * C, H, W, _p(y), _stream()), "dp_subsample2")
    return y


def subsample2_add_(g, dy):
    """g[:, :, ::2, ::2] += dy, in place: the adjoint of subsample2 accumulated into an existing gradient."""
    lib = _lib.load()
    _chk(g, torch.float32, "g"), _chk(dy, torch.float32, "dy")
    N, C, H, W = g.shape
    assert dy.shape == (N, C, H // 2, W // 2)
    _lib.check(lib.dp_subsample2_add(_p(dy), N * C, H, W, _p(g), _stream()), "dp_subsample2_add")
    return g


class DualConv1x1Function(torch.autograd.Function):
    """The two frozen 1x1 convolutions that read the pre-activation of a bottleneck's first block —
    ``conv1`` (stride 1) and ``downsample.conv`` (stride ``s`` in {1, 2}) — as ONE autograd node:

    forward   branch = conv1x1(pre, w1);  shortcut = conv1x1(subsample_s(pre), wd)
    backward  g = conv1x1_bwd(d_branch, w1);  g[even pixels] += conv1x1_bwd(d_shortcut, wd)   (in place)

    Replaces, per first block of stages 1-3: MIOpen's NCHW<->NHWC transposes of the FULL-resolution
    activation around its strided kernel, the zero-fill + scatter of the full-resolution gradient, and
    autograd's accumulation add of the two branch gradients.  The arithmetic is the
    reference's: a stride-2 1x1 convolution reads exactly the even pixels."""

    @staticmethod
    def forward(ctx, pre, w1, wd, stride):
        from . import conv1x1
        pre = pre.contiguous()
        branch = conv1x1._run("fwd", pre, w1)
        small = pre if stride == 1 else subsample2(pre)
        shortcut = conv1x1._run("fwd", small, wd)
        # `pre` / `small`: shape references for MIOpen's backward-data only, never re-read
        ctx.save_for_backward(w1, wd, pre, small if stride != 1 else pre)
        ctx.stride = stride
        ctx.set_materialize_grads(False)
        return branch, shortcut

    @staticmethod
    def backward(ctx, d_branch, d_short):
        from . import conv1x1
        w1, wd, pre, small = ctx.saved_tensors
        g = None
        if d_branch is not None:
            g = conv1x1._run("bwd", d_branch.contiguous(), w1, pre)
        if d_short is not None:
            d_short = d_short.contiguous()
            if ctx.stride == 1:
                if g is None:
                    g = conv1x1._run("bwd", d_short, wd, pre)
                else:       # accumulate inside the GEMM (beta = 1): no separate add pass
                    N, O, H, W = d_short.shape
                    C = wd.shape[1]
                    g.view(N, C, H * W).baddbmm_(wd.view(O, C).t().unsqueeze(0).expand(N, C, O),
                                                 d_short.view(N, O, H * W))
            else:
                gs = conv1x1._run("bwd", d_short, wd, small)
                if g is None:
                    g = torch.zeros_like(pre)
                subsample2_add_(g, gs)
        return g, None, None, None


# ---------------------------------------------------------------- a-8 (round 5): GroupNorm folded into the consuming convolution
def _fconv_fwd(kind, x, w, ab, res):
    from . import libconv
    if kind == 1:
        return conv1x1_fwd(x, libconv.packed1(w, False), ab=ab, res=res)
    assert res is None
    if kind == 32:          # 3x3 / stride 2
        return conv3x3s2_fwd(x, libconv._packed3(w, False), ab=ab)
    return libconv.conv3x3_s1(x, w, False, ab=ab)


def _fconv_bwd(kind, dy, w, res=None, x_ref=None):
    from . import libconv
    if kind == 1:
        return conv1x1_fwd(dy, libconv.packed1(w, True), res=res, out=res)
    assert res is None
    if kind == 32:          # dp_conv3x3s2_bwd where libconv routes it there, else the library (x_ref: its shape only)
        return libconv.conv_bwd_data(dy, x_ref, w, (2, 2), (1, 1))
    return libconv.conv3x3_s1(dy, w, True)


def gn_fold_supported(x, conv_weight, groups, stride=(1, 1), padding=None):
    """Can ``conv(relu(group_norm(x)))`` run as dp_gn_stats + a convolution that applies the norm while staging?
    1x1 / stride 1: any plane of H*W % 4 == 0 pixels; 3x3 / stride 1 / pad 1: square planes of side 56 / 28 / 14;
    3x3 / stride 2 / pad 1: the same sides (dp_conv3x3s2_fwd; forward only — its input gradient is the library's)."""
    if not gn_relu_supported(x, groups) or x.dim() != 4:
        return False
    k = tuple(conv_weight.shape[2:])
    if k == (1, 1):
        return tuple(stride) == (1, 1) and (x.shape[2] * x.shape[3]) % 4 == 0 and conv1x1_supported(x, conv_weight)
    if k == (3, 3):
        padding = (1, 1) if padding is None else padding
        if tuple(stride) == (2, 2):
            from . import libconv
            return (libconv.CONV3X3S2 != "off" and conv3x3s2_supported(x, conv_weight, stride, padding)
                    and libconv.conv3x3s2_fwd_pays(x, conv_weight))
        return conv3x3_supported(x, conv_weight, stride, padding) and x.shape[2] in CONV3X3_FOLD_SIDES
    return False


class GnConvFunction(torch.autograd.Function):
    """``out = conv(relu(group_norm(x)), w) [+ add]`` for a frozen 1x1 (``kind`` 1), 3x3 (``kind`` 3) stride-1 or 3x3
    stride-2 (``kind`` 32: own forward, the library's input gradient) filter,
    the GroupNorm-apply + ReLU folded into the convolution's operand staging: one statistics pass over x
    (dp_gn_stats), then dp_conv1x1_fwd / dp_conv3x3_gn_fwd read the RAW x — the normalised activation is never written or
    re-read.  ``add`` (1x1 only): the residual, added in the convolution's epilogue.  ``passthrough``: also return x itself
    (as a second output whose gradient comes back into this node's GroupNorm backward as ``dres``: the shortcut of a
    non-first bottleneck — replaces autograd's accumulation add).  Bit-identical to GnReluFunction + the plain kernel.
    Backward: input gradient of the convolution (the same MFMA kernel on the transposed weights), then dp_gn_relu_bwd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, w, kind, add, passthrough):
        x = x.contiguous()
        mean, rstd, ab, _ = gn_stats(x, gamma, beta, groups, eps)
        out = _fconv_fwd(kind, x, w, ab, None if add is None else add.contiguous())
        ctx.save_for_backward(x, gamma, beta, mean, rstd, w)
        ctx.groups, ctx.kind, ctx.has_add, ctx.passthrough = groups, kind, add is not None, passthrough
        ctx.set_materialize_grads(False)
        return (x, out) if passthrough else out

    @staticmethod
    def backward(ctx, *grads):
        x, gamma, beta, mean, rstd, w = ctx.saved_tensors
        d_pass, d_out = grads if ctx.passthrough else (None, grads[0])
        dx = d_pass
        if d_out is not None:
            d_out = d_out.contiguous()
            dy = _fconv_bwd(ctx.kind, d_out, w, x_ref=x)
            dx = gn_relu_bwd(dy, x, gamma, beta, mean, rstd, ctx.groups, dres=None if d_pass is None else d_pass.contiguous())
        return dx, None, None, None, None, None, None, (d_out if ctx.has_add else None), None


class GnDualConvFunction(torch.autograd.Function):
    """First block of a stage, folded: ``branch = conv1x1(y, w1)``, ``shortcut = conv1x1(subsample_s(y), wd)`` with
    ``y = relu(group_norm(x))`` never materialised (both convolutions apply the norm while staging; the coefficients are per
    (sample, channel), so the subsampled RAW tensor takes the same table).  Backward as DualConv1x1Function (the stride-1
    downsample accumulates in the kernel's epilogue), then dp_gn_relu_bwd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, w1, wd, stride):
        x = x.contiguous()
        mean, rstd, ab, _ = gn_stats(x, gamma, beta, groups, eps)
        branch = _fconv_fwd(1, x, w1, ab, None)
        small = x if stride == 1 else subsample2(x)
        shortcut = _fconv_fwd(1, small, wd, ab, None)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, w1, wd)
        ctx.groups, ctx.stride = groups, stride
        ctx.set_materialize_grads(False)
        return branch, shortcut

    @staticmethod
    def backward(ctx, d_branch, d_short):
        x, gamma, beta, mean, rstd, w1, wd = ctx.saved_tensors
        g = None
        if d_branch is not None:
            g = _fconv_bwd(1, d_branch.contiguous(), w1)
        if d_short is not None:
            d_short = d_short.contiguous()
            if ctx.stride == 1:
                g = _fconv_bwd(1, d_short, wd, res=g)          # g += W_d^T d_short in the epilogue (plain when g is None)
            else:
                gs = _fconv_bwd(1, d_short, wd)
                if g is None:
                    g = torch.zeros_like(x)
                subsample2_add_(g, gs)
        if g is None:
            return (None,) * 8
        return gn_relu_bwd(g, x, gamma, beta, mean, rstd, ctx.groups), None, None, None, None, None, None, None


class GnReluPassFunction(torch.autograd.Function):
    """``x -> (x, relu(group_norm(x)))``: GnReluFunction that also hands x on (the shortcut of a non-first bottleneck) and
    takes the gradient arriving there as ``dres`` of dp_gn_relu_bwd instead of leaving an accumulation add to autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        x = x.contiguous()
        y, mean, rstd, _ = gn_relu_fwd(x, weight, bias, groups, eps)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.groups = groups
        ctx.set_materialize_grads(False)
        return x, y

    @staticmethod
    def backward(ctx, d_pass, dy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        if dy is None:
            return d_pass, None, None, None, None
        dx = gn_relu_bwd(dy.contiguous(), x, weight, bias, mean, rstd, ctx.groups,
                         dres=None if d_pass is None else d_pass.contiguous())
        return dx, None, None, None, None


class Conv1x1AddFunction(torch.autograd.Function):
    """``conv1x1(y, w) + add`` on dp_conv1x1_fwd with the add in the epilogue (the 7 x 7 planes, where the GroupNorm fold
    does not apply)."""

    @staticmethod
    def forward(ctx, y, w, add):
        ctx.save_for_backward(w)
        return _fconv_fwd(1, y.contiguous(), w, None, add.contiguous())

    @staticmethod
    def backward(ctx, d_out):
        (w,) = ctx.saved_tensors
        d_out = d_out.contiguous()
        return _fconv_bwd(1, d_out, w), None, d_out
