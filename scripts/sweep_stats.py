#!/usr/bin/env python
"""Per-category kernel time of the collect_failure sweep from a rocprofv3 kernel trace of
`bench.py --steps 1 --warmup 1` (the sweep is everything after the last k_project_update launch)."""
import csv
import sys
from collections import defaultdict


def cat(n):
    if "k_gn_relu" in n:
        return "GroupNorm+ReLU forward (ours)"
    if "k_apply_fwd" in n:
        return "dp_apply_fwd (ours)"
    if "k_argmax" in n:
        return "dp_argmax (ours)"
    if "k_pad_maxpool" in n:
        return "pad+maxpool forward (ours)"
    if "k_subsample2" in n:
        return "subsample (ours)"
    if "miopenSp3AsmConv" in n:
        return "MIOpen Winograd"
    if n.startswith("igemm"):
        return "MIOpen igemm NHWC"
    if "Cijk" in n:
        return "Tensile / hipBLASLt / rocBLAS GEMM"
    if "transpose" in n or "SubTensorOp" in n:
        return "MIOpen transposes / zero-fill"
    return "other (torch eager: where, compare, mean, ...)"


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    last_update = max(i for i, r in enumerate(rows) if "k_project_update" in r[0])
    sel = rows[last_update + 1:]
    t0, t1 = sel[0][1], max(r[2] for r in sel)
    agg = defaultdict(float)
    for n, s, e in sel:
        agg[cat(n)] += (e - s) / 1e6
    tot = sum(agg.values())
    print("# collect_failure sweep: %.1f ms wall, %.1f ms of kernel time (%.1f %% busy), %d dispatches"
          % ((t1 - t0) / 1e6, tot, 100 * tot / ((t1 - t0) / 1e6), len(sel)))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
        print("%-52s %10.1f ms %6.1f %%" % (k, v, 100 * v / tot))


if __name__ == "__main__":
    main()
