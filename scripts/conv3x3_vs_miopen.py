"""Go / no-go measurement: dp_conv3x3_fwd (direct implicit GEMM on v_mfma_f32_32x32x2_f32) against
MIOpen's route for the same convolution, on the GPU box, same process, same tensors.

    python scripts/conv3x3_vs_miopen.py [N ...]    # default 512 = the headline micro-batch

Prints one JSON line per (batch, shape) — the four stride-1 3x3 convolutions of ResNetV2-50 at 224 x 224: ms and effective
TFLOP/s of both routes, forward and input gradient (= the same kernel on transposed + flipped weights), and the max abs
difference relative to the output scale.  The route table dorpatch_amd/conv3x3_gfx950.json is derived from this output.
``--stride2``: the three stride-2 3x3 convolutions on dp_conv3x3s2_fwd / dp_conv3x3s2_bwd instead (round 5).
``--flat``: the stride-1 shapes on both kernels behind dp_conv3x3_fwd (k_conv3x3_mfma / k_conv3x3_flat, DP_DEBUG_CONV3X3_VARIANT)
against MIOpen; ``--384``: the planes of a 384 x 384 input (96 / 48 / 24 / 12: flat kernel only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from dorpatch_amd import ops


def timed(fn, iters=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


SHAPES = ((64, 56), (128, 28), (256, 14), (512, 7))      # (channels, side) of ResNetV2-50's stride-1 3x3 convolutions @224


def one(N, C, S):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, S, S, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).cuda()
    wt, wt_bwd = ops.pack_conv3x3_weights(w), ops.pack_conv3x3_weights(w, transpose=True)
    flop = 2.0 * N * S * S * C * C * 9
    want = F.conv2d(x, w, padding=1)
    got = ops.conv3x3_fwd(x, wt)
    err = float((got - want).abs().max() / want.abs().max())
    dy = torch.randn(N, C, S, S, generator=g).cuda()
    bwd_lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                          (True, False, False))[0]
    want_b, got_b = bwd_lib(), ops.conv3x3_fwd(dy, wt_bwd)
    err_b = float((got_b - want_b).abs().max() / want_b.abs().max())
    ms = dict(miopen_fwd=timed(lambda: F.conv2d(x, w, padding=1)), mfma_fwd=timed(lambda: ops.conv3x3_fwd(x, wt)),
              miopen_bwd_data=timed(bwd_lib), mfma_bwd_data=timed(lambda: ops.conv3x3_fwd(dy, wt_bwd)))
    return dict(shape="N=%d %d->%d 3x3/1 @%dx%d fp32" % (N, C, C, S, S), gflop=round(flop / 1e9, 2),
                ms={k: round(v, 4) for k, v in ms.items()},
                tflops_effective={k: round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items()},
                speedup=dict(fwd=round(ms["miopen_fwd"] / ms["mfma_fwd"], 3), bwd_data=round(ms["miopen_bwd_data"] / ms["mfma_bwd_data"], 3)),
                max_rel_diff_fwd=err, max_rel_diff_bwd_data=err_b)


SHAPES_S2 = ((128, 56), (256, 28), (512, 14))      # (channels, INPUT side) of the three stride-2 3x3 convolutions @224


def one_s2(N, C, S):
    """dp_conv3x3s2_fwd (round 5) against F.conv2d(stride 2): plain, and with the GroupNorm + ReLU in front of it folded
    in (against dp_gn_relu_fwd + F.conv2d = what the round-4 graph runs)."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, S, S, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).cuda()
    wt = ops.pack_conv3x3_weights(w)
    gamma, beta = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda() * 0.2
    flop = 2.0 * N * (S // 2) ** 2 * C * C * 9
    want = F.conv2d(x, w, stride=2, padding=1)
    err = float((ops.conv3x3s2_fwd(x, wt) - want).abs().max() / want.abs().max())

    def lib_gn():
        return F.conv2d(ops.gn_relu_fwd(x, gamma, beta, 32, 1e-5)[0], w, stride=2, padding=1)

    def own_gn():
        return ops.conv3x3s2_fwd(x, wt, ab=ops.gn_stats(x, gamma, beta, 32, 1e-5)[2])

    err_gn = float((own_gn() - lib_gn()).abs().max() / want.abs().max())
    dy = torch.randn(N, C, S // 2, S // 2, generator=g).cuda()
    wtb = ops.pack_conv3x3s2_dgrad_weights(w)
    bwd_lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                                          (True, False, False))[0]
    want_b, got_b = bwd_lib(), ops.conv3x3s2_bwd(dy, wtb, C)
    err_b = float((got_b - want_b).abs().max() / want_b.abs().max())
    ms = dict(miopen_fwd=timed(lambda: F.conv2d(x, w, stride=2, padding=1)), mfma_fwd=timed(lambda: ops.conv3x3s2_fwd(x, wt)),
              gn_then_miopen_fwd=timed(lib_gn), stats_then_mfma_fold_fwd=timed(own_gn),
              miopen_bwd_data=timed(bwd_lib), mfma_bwd_data=timed(lambda: ops.conv3x3s2_bwd(dy, wtb, C)))
    return dict(shape="N=%d %d->%d 3x3/2 @%dx%d fp32" % (N, C, C, S, S), gflop=round(flop / 1e9, 2),
                ms={k: round(v, 4) for k, v in ms.items()},
                tflops_effective={k: round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items()},
                speedup=dict(fwd=round(ms["miopen_fwd"] / ms["mfma_fwd"], 3),
                             gn_fwd=round(ms["gn_then_miopen_fwd"] / ms["stats_then_mfma_fold_fwd"], 3),
                             bwd_data=round(ms["miopen_bwd_data"] / ms["mfma_bwd_data"], 3)),
                max_rel_diff_fwd=err, max_rel_diff_gn_fwd=err_gn, max_rel_diff_bwd_data=err_b)


def one_wino(N, C, S):
    """Round 6: dp_conv3x3_wino_fwd (Winograd F(2x2, 3x3), position GEMMs on the matrix cores) against the direct kernel
    (dp_conv3x3_fwd) and MIOpen: forward, input gradient, and forward with the GroupNorm fold (fed the same coefficients)."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, S, S, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).cuda()
    wt, wt_bwd = ops.pack_conv3x3_weights(w), ops.pack_conv3x3_weights(w, transpose=True)
    ww, ww_bwd = ops.pack_conv3x3_wino_weights(w), ops.pack_conv3x3_wino_weights(w, transpose=True)
    gamma, beta = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda() * 0.2
    flop = 2.0 * N * S * S * C * C * 9
    want = F.conv2d(x.double(), w.double(), padding=1)
    scale = float(want.abs().max())
    err_w = float((ops.conv3x3_wino_fwd(x, ww).double() - want).abs().max()) / scale
    err_d = float((ops.conv3x3_fwd(x, wt).double() - want).abs().max()) / scale
    dy = torch.randn(N, C, S, S, generator=g).cuda()
    bwd_lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                          (True, False, False))[0]
    ms = dict(miopen_fwd=timed(lambda: F.conv2d(x, w, padding=1)), direct_fwd=timed(lambda: ops.conv3x3_fwd(x, wt)),
              wino_fwd=timed(lambda: ops.conv3x3_wino_fwd(x, ww)),
              miopen_bwd_data=timed(bwd_lib), direct_bwd_data=timed(lambda: ops.conv3x3_fwd(dy, wt_bwd)),
              wino_bwd_data=timed(lambda: ops.conv3x3_wino_fwd(dy, ww_bwd)))
    if S != 7:
        ab = ops.gn_stats(x, gamma, beta, 32, 1e-5)[2]
        ms["direct_fold_fwd"] = timed(lambda: ops.conv3x3_fwd(x, wt, ab=ab))
        ms["wino_fold_fwd"] = timed(lambda: ops.conv3x3_wino_fwd(x, ww, ab=ab))
    return dict(shape="N=%d %d->%d 3x3/1 @%dx%d fp32" % (N, C, C, S, S), gflop=round(flop / 1e9, 2),
                ms={k: round(v, 4) for k, v in ms.items()},
                tflops_effective={k: round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items()},
                speedup_over_direct=dict(fwd=round(ms["direct_fwd"] / ms["wino_fwd"], 3),
                                         bwd_data=round(ms["direct_bwd_data"] / ms["wino_bwd_data"], 3)),
                max_err_vs_fp64_of_scale=dict(wino=err_w, direct=err_d))


SHAPES_384 = ((64, 96), (128, 48), (256, 24), (512, 12))


def one_flat(N, C, S):
    """dp_conv3x3_fwd on k_conv3x3_mfma (where it takes the side) and on k_conv3x3_flat against MIOpen, forward and input
    gradient, plain and with the GroupNorm fold (fold vs dp_gn_relu_fwd + the same kernel)."""
    from dorpatch_amd import _lib
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, S, S, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).cuda()
    wt, wt_bwd = ops.pack_conv3x3_weights(w), ops.pack_conv3x3_weights(w, transpose=True)
    gamma, beta = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda() * 0.2
    flop = 2.0 * N * S * S * C * C * 9
    want = F.conv2d(x, w, padding=1)
    dy = torch.randn(N, C, S, S, generator=g).cuda()
    bwd_lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                          (True, False, False))[0]
    ms = dict(miopen_fwd=timed(lambda: F.conv2d(x, w, padding=1)), miopen_bwd_data=timed(bwd_lib))
    out = {}
    for name, variant in (("rows", 1), ("flat", 2)):
        if variant == 1 and S not in (56, 28, 14, 7):
            continue
        ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, variant)
        try:
            out[name] = ops.conv3x3_fwd(x, wt)
            ms[name + "_fwd"] = timed(lambda: ops.conv3x3_fwd(x, wt))
            ms[name + "_bwd_data"] = timed(lambda: ops.conv3x3_fwd(dy, wt_bwd))
            if S != 7:
                ms[name + "_stats_fold_fwd"] = timed(lambda: ops.conv3x3_fwd(x, wt, ab=ops.gn_stats(x, gamma, beta, 32, 1e-5)[2]))
        finally:
            ops.debug_set(_lib.DP_DEBUG_CONV3X3_VARIANT, 0)
    if S != 7:
        ms["gn_then_miopen_fwd"] = timed(lambda: F.conv2d(ops.gn_relu_fwd(x, gamma, beta, 32, 1e-5)[0], w, padding=1))
    return dict(shape="N=%d %d->%d 3x3/1 @%dx%d fp32" % (N, C, C, S, S), gflop=round(flop / 1e9, 2),
                ms={k: round(v, 4) for k, v in ms.items()},
                tflops_effective={k: round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items()},
                flat_equals_rows=(torch.equal(out["flat"], out["rows"]) if "rows" in out else None),
                max_rel_diff_fwd=float((out["flat"] - want).abs().max() / want.abs().max()))


def main():
    torch.backends.cudnn.benchmark = False          # the product's setting: MIOpen immediate mode
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batches = [int(a) for a in args] or [512]
    for N in batches:
        if "--flat" in sys.argv:
            for C, S in (SHAPES_384 if "--384" in sys.argv else SHAPES):
                print(json.dumps(one_flat(N, C, S)), flush=True)
            continue
        if "--wino" in sys.argv:
            for C, S in (SHAPES_384 if "--384" in sys.argv else SHAPES):
                print(json.dumps(one_wino(N, C, S)), flush=True)
            continue
        if "--stride2" in sys.argv:
            for C, S in SHAPES_S2:
                print(json.dumps(one_s2(N, C, S)), flush=True)
            continue
        for C, S in SHAPES:
            print(json.dumps(one(N, C, S)), flush=True)


if __name__ == "__main__":
    main()
