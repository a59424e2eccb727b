"""Go / no-go measurement per shape: dp_conv1x1_fwd (NCHW-in-place GEMM on v_mfma_f32_32x32x2_f32)
against the library routes of the same 1x1 convolution — the batched GEMM with the TUNED solutions of
tunableop_gfx950.csv (what the product ran until round 4) and MIOpen — on the GPU box, same process, same tensors.

    python scripts/conv1x1_vs_lib.py [--variants 0,8,1,2,4] [N ...]      # default N = 512, the headline micro-batch

One JSON line per (batch, direction, C, O, HW): ms of every route, effective TFLOP/s, speed-up over the best library route,
max abs difference relative to the output scale.  `--variants`: dp_debug_set(DP_DEBUG_CONV1X1_VARIANT) values to time
(0 = the product's launch).  The route table dorpatch_amd/conv1x1_gfx950.json gets its "mfma" entries from this output."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from dorpatch_amd import _lib, conv1x1, ops

# (C, O, side) of ResNetV2-50's stride-1 1x1 convolutions at 224 x 224 (incl. the subsampled downsample convolutions)
SHAPES = ((64, 64, 56), (64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (256, 512, 28), (512, 128, 28),
          (512, 256, 28), (256, 1024, 14), (512, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7),
          (1024, 2048, 7), (2048, 512, 7))


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


def one(N, direction, C, O, S, variants):
    g = torch.Generator().manual_seed(C + O)
    w = (torch.randn(O, C, 1, 1, generator=g) / C ** 0.5).cuda()
    x = torch.randn(N, C, S, S, generator=g).cuda()
    t = x if direction == "fwd" else torch.randn(N, O, S, S, generator=g).cuda()
    flop = 2.0 * N * S * S * C * O
    ms, err = {}, None
    want = conv1x1._IMPL[(direction, "gemm")](t, w, x)
    for route in ("gemm", "miopen"):
        ms[route] = timed(lambda: conv1x1._IMPL[(direction, route)](t, w, x))
    for v in variants:
        ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, v)
        got = conv1x1._IMPL[(direction, "mfma")](t, w, x)
        e = float((got - want).abs().max() / want.abs().max())
        err = e if err is None else max(err, e)
        ms["mfma" if v == 0 else "mfma_v%d" % v] = timed(lambda: conv1x1._IMPL[(direction, "mfma")](t, w, x))
    ops.debug_set(_lib.DP_DEBUG_CONV1X1_VARIANT, 0)
    lib = min(ms["gemm"], ms["miopen"])
    best = min(v for k, v in ms.items() if k.startswith("mfma"))
    return dict(N=N, dir=direction, C=C, O=O, HW=S * S, gflop=round(flop / 1e9, 2), ms={k: round(v, 4) for k, v in ms.items()},
                tflops={k: round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items()},
                lib_over_mfma=round(lib / ms.get("mfma", best), 3), lib_over_best_variant=round(lib / best, 3), max_rel_diff=err)


def main():
    torch.backends.cudnn.benchmark = False          # the product's setting: MIOpen immediate mode
    args = sys.argv[1:]
    variants = [0]
    if args and args[0] == "--variants":
        variants = [int(v) for v in args[1].split(",")]
        args = args[2:]
    batches = [int(a) for a in args] or [512]
    tuned = conv1x1.activate(None, True)             # the tuned GEMM solutions, as inside generate()
    print(json.dumps(dict(tuned_gemm_solutions=bool(tuned), report=conv1x1.report_tuned())), flush=True)
    for N in batches:
        for C, O, S in SHAPES:
            for direction in ("fwd", "bwd"):
                print(json.dumps(one(N, direction, C, O, S, variants)), flush=True)
    if tuned:
        conv1x1.deactivate()


if __name__ == "__main__":
    main()
