#!/usr/bin/env python
"""Times both library routes of every frozen 1x1/1 convolution shape of ResNetV2-50 at one micro-batch
(dorpatch_amd/conv1x1.py: in-place NCHW batched GEMM vs MIOpen), forward and input-gradient, and prints
one JSON line per (direction, shape) as soon as it is measured (most expensive shapes first).
    python scripts/conv1x1_table.py [--n 512] [--size 224]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dorpatch_amd import conv1x1  # noqa: E402


def shapes(size):
    """(C_in, C_out, HW) of every stride-1 1x1 convolution, with its multiplicity per forward."""
    out = {}
    hw = (size // 4) ** 2
    prev = 64
    for s, (depth, ch) in enumerate(zip((3, 4, 6, 3), (256, 512, 1024, 2048))):
        mid = ch // 4
        hw_out = hw if s == 0 else hw // 4
        for b in range(depth):
            cin = prev if b == 0 else ch
            hw_in = hw if b == 0 else hw_out
            if b == 0 and s == 0:
                out[(cin, ch, hw_in)] = out.get((cin, ch, hw_in), 0) + 1          # stride-1 downsample conv
            out[(cin, mid, hw_in)] = out.get((cin, mid, hw_in), 0) + 1            # conv1
            out[(mid, ch, hw_out)] = out.get((mid, ch, hw_out), 0) + 1            # conv3
        prev, hw = ch, hw_out
    return out


def downsample_shapes(size):
    """The strided downsample convolutions as DualConv1x1Function runs them: stride-1 1x1 on the subsampled
    pre-activation, (C_in, C_out, HW_out) for stages 1-3."""
    hw = (size // 4) ** 2
    return {(256, 512, hw // 4): 1, (512, 1024, hw // 16): 1, (1024, 2048, hw // 64): 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--find", type=int, default=0, help="1: MIOpen exhaustive find for the MIOpen route")
    ap.add_argument("--blas", default="default", help="torch.backends.cuda.preferred_blas_library: default | hipblaslt | cublas (= rocBLAS)")
    ap.add_argument("--only-downsample", action="store_true")
    args = ap.parse_args()
    if args.blas != "default":
        torch.backends.cuda.preferred_blas_library(args.blas)
    torch.backends.cudnn.benchmark = bool(args.find)
    conv1x1.MODE = "auto"
    dev = torch.device("cuda", 0)
    todo = {} if args.only_downsample else dict(shapes(args.size))
    for k, v in downsample_shapes(args.size).items():
        todo.setdefault(k, v)
    todo = sorted(todo.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2])
    for (C, O, HW), mult in todo:
        H = int(round(HW ** 0.5))
        w = torch.randn(O, C, 1, 1, device=dev) / C ** 0.5
        x = torch.randn(args.n, C, H, H, device=dev)
        dy = torch.randn(args.n, O, H, H, device=dev)
        for direction, t in (("fwd", x), ("bwd", dy)):
            conv1x1._pick(direction, t, w, x)
            key = (direction, args.n, C, O, HW)
            g, m = conv1x1._timings[key]
            flop = 2.0 * args.n * C * O * HW
            print(json.dumps({"dir": direction, "N": args.n, "C": C, "O": O, "HW": HW, "per_forward": mult,
                              "gemm_ms": round(g, 4), "miopen_ms": round(m, 4), "choice": conv1x1._choice[key],
                              "gemm_TFLOPs": round(flop / g / 1e9, 1), "miopen_TFLOPs": round(flop / m / 1e9, 1)}),
                  flush=True)
        del x, dy, w


if __name__ == "__main__":
    main()
