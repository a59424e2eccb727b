#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel
stats table restricted to the bench's TIMED window (so warm-up / MIOpen solver search do not pollute it).

    python scripts/rocpd_stats.py gpurun_out/prof/bench_results.db --timed-steps 2 > profiles/r01_....txt

The timed window starts at the k_sumsq_partials launch that opens the first timed step (the
`--timed-steps`-th last full-size k_apply_fwd marks that step) and ends at the last kernel.
"""
import argparse
import csv
import sqlite3
import sys
from collections import defaultdict


def load(path):
    """-> list of (name, start_ns, end_ns, grid_z)"""
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        return [(r[0], r[1], r[2], r[3]) for r in
                c.execute("select name, start, end, grid_z from kernels order by start")]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                         int(r.get("Grid_Size_Z", r.get("Grid_Size_z", 1)) or 1)))
    rows.sort(key=lambda r: r[1])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--timed-steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--all", action="store_true", help="whole trace instead of the timed window")
    args = ap.parse_args()
    rows = load(args.path)
    t_begin = rows[0][1]
    if not args.all:
        applies = [r for r in rows if "k_apply_fwd" in r[0] or "k_apply_affine_fwd" in r[0]]   # the latter: bench.py --placement
        zmax = max(r[3] for r in applies)
        full = [r for r in applies if r[3] == zmax]
        first_timed = full[-args.timed_steps][1]
        opens = [r[1] for r in rows if "k_sumsq_partials" in r[0] and r[1] < first_timed]
        t_begin = opens[-1]
    sel = [r for r in rows if r[1] >= t_begin]
    t_end = max(r[2] for r in sel)
    agg = defaultdict(lambda: [0, 0, 1 << 62, 0])
    for name, s, e, _ in sel:
        a = agg[name]
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    total = sum(a[1] for a in agg.values())
    print("# source: %s" % args.path)
    print("# window: %.3f ms wall, %.3f ms of kernel time (%.1f%% busy), %d dispatches, %d timed steps"
          % ((t_end - t_begin) / 1e6, total / 1e6, 100.0 * total / (t_end - t_begin), len(sel), args.timed_steps))
    print("%-96s %7s %11s %6s %11s %10s %10s" % ("kernel", "calls", "total_ms", "%", "avg_us", "min_us", "max_us"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print("%-96s %7d %11.3f %6.2f %11.2f %10.2f %10.2f" % (name[:96], a[0], a[1] / 1e6, 100.0 * a[1] / total,
                                                                a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3))
    ours = {k: v for k, v in agg.items() if "(anonymous namespace)::k_" in k or k.startswith("k_")}
    if ours:
        print("# --- dorpatch_hip kernels")
        for name, a in sorted(ours.items(), key=lambda kv: -kv[1][1]):
            print("%-96s %7d %11.3f %6.2f %11.2f %10.2f %10.2f" % (name[:96], a[0], a[1] / 1e6, 100.0 * a[1] / total,
                                                                    a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3))


if __name__ == "__main__":
    sys.exit(main())
