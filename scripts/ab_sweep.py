#!/usr/bin/env python
"""A/B of host-side knobs of the hot loop in ONE process (model built once): micro-batch size x the
library route of the frozen 1x1 convolutions.  Prints one JSON line per configuration:
    python scripts/ab_sweep.py --micro-batches 128,256,512,1024 --modes auto,miopen --steps 3
Same workload and step function as bench.py (BASELINE configs[1]); no CPU baseline, no sweep."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--micro-batches", default="128,256,512,1024")
    ap.add_argument("--modes", default="auto,miopen")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--size", type=int, default=224)
    args = ap.parse_args()
    import bench
    from dorpatch_amd import conv1x1
    from dorpatch_amd.attack import DorPatch, HotLoop
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    model = bench.build_model(dev)
    B, S, H = args.batch, args.samples, args.size
    x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1234)).to(dev)
    y = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(7)).to(dev)
    for mode_spec in args.modes.split(","):
        mode, _, blas = mode_spec.partition("@")          # e.g. auto@hipblaslt, auto@cublas (= rocBLAS)
        try:
            torch.backends.cuda.preferred_blas_library(blas or "default")
        except Exception as e:                            # noqa: BLE001 - report and carry on with the default
            print(json.dumps({"conv1x1": mode_spec, "skipped": repr(e)}), flush=True)
            continue
        conv1x1.reset()
        for mb in [int(v) for v in args.micro_batches.split(",")]:
            conv1x1.MODE = mode
            np.random.seed(1234)
            loop = HotLoop(DorPatch(micro_batch=mb, verbose=False), model, x, 0.0204, 1000, "ab_out/cfg/sub", 0, y, True,
                           1e-2, 1e-1, 0, 1, 10 ** 9, 7, 'topk', 2, S, 1e-3, 1e-3, 4.0, False, {})
            loop.step(1)                              # warm-up (+ per-shape calibration in auto mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                loop.step(2 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            loop.close()
            print(json.dumps({"conv1x1": mode_spec, "micro_batch": mb, "ms_per_step": round(dt * 1e3, 2),
                              "eot_samples_per_s": round(B * S / dt, 1),
                              "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                              "conv1x1_report": conv1x1.report() if mode == "auto" else None}), flush=True)
            torch.cuda.reset_peak_memory_stats()
        if mode != "auto":
            continue
        rows = [{"dir": k[0], "N": k[1], "C": k[2], "O": k[3], "HW": k[4], "gemm_ms": round(v[0], 4),
                 "miopen_ms": round(v[1], 4)} for k, v in sorted(conv1x1._timings.items())]
        print(json.dumps({"conv1x1_calibration": mode_spec, "rows": rows}), flush=True)


if __name__ == "__main__":
    main()
