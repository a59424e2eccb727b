#!/usr/bin/env python
"""Can PyTorch's TunableOp find faster hipBLASLt / rocBLAS solutions for the batched GEMMs of the conv1x1 GEMM route?
For every (direction, shape) of the committed table: time the route with the default solution, then let TunableOp
tune that GEMM (bounded) and time again.  Prints one JSON line per shape + a summary; writes the TunableOp CSV next to
its output.  Diagnostic (decides whether a tuned-solution file is worth committing)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dorpatch_amd import conv1x1  # noqa: E402


def time_ms(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--size", type=int, default=224, help="input side: 224 or 384 (the layer shapes follow from it)")
    ap.add_argument("--csv", default="tunableop_gfx950.csv")
    ap.add_argument("--max-ms", type=int, default=300)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from scripts.conv1x1_table import downsample_shapes, shapes as stride1_shapes
    todo = set(stride1_shapes(args.size)) | set(downsample_shapes(args.size))
    shapes = sorted((d, C, O, HW) for (C, O, HW) in todo for d in ("fwd", "bwd"))
    ops = []
    for (direction, C, O, HW) in shapes:
        H = int(round(HW ** 0.5))
        w = torch.randn(O, C, 1, 1, device=dev) / C ** 0.5
        t = torch.randn(args.n, C if direction == "fwd" else O, H, H, device=dev)
        ops.append((direction, C, O, HW, w, t))
    base, lib = {}, {}
    for direction, C, O, HW, w, t in ops:
        base[(direction, C, O, HW)] = time_ms(lambda: conv1x1._IMPL[(direction, "gemm")](t, w, None))
        H = int(round(HW ** 0.5))
        xref = t if direction == "fwd" else torch.empty(args.n, C, H, H, device=dev)
        lib[(direction, C, O, HW)] = time_ms(lambda: conv1x1._IMPL[(direction, "miopen")](t, w, xref))
        del xref
    import torch.cuda.tunable as tun
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_duration(args.max_ms)
    tun.set_max_tuning_iterations(args.iters)
    tun.set_filename(args.csv)
    tot_b = tot_t = 0.0
    for direction, C, O, HW, w, t in ops:
        fn = lambda: conv1x1._IMPL[(direction, "gemm")](t, w, None)
        fn()                                   # tunes this GEMM
        tuned = time_ms(fn)
        b = base[(direction, C, O, HW)]
        route = conv1x1.TABLE.get((direction, C, O, HW), "miopen")
        print(json.dumps(dict(dir=direction, N=args.n, size=args.size, C=C, O=O, HW=HW, table_route=route, gemm_default_ms=round(b, 4),
                              gemm_tuned_ms=round(tuned, 4), speedup=round(b / tuned, 3),
                              miopen_ms=round(lib[(direction, C, O, HW)], 4))), flush=True)
        if route == "gemm":
            tot_b += b
            tot_t += tuned
    tun.write_file(args.csv) if hasattr(tun, "write_file") else None
    print(json.dumps(dict(summary="GEMM-routed shapes (one call each)", default_ms=round(tot_b, 3), tuned_ms=round(tot_t, 3),
                          csv=args.csv)), flush=True)


if __name__ == "__main__":
    main()
