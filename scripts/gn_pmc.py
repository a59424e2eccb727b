#!/usr/bin/env python
"""HBM traffic of the GroupNorm+ReLU kernels (dp_gn_relu_fwd / _bwd, the largest hand-written share of a step) from
rocprofv3 PMC counters, next to their algorithmic bytes.

    python scripts/gn_pmc.py [out.json]

Runs `tools/kbench 64 32 224 2 dp_gn_relu` (the 5 layer shapes of ResNetV2-50 at 256 samples x {fwd, fwd+res, bwd,
bwd+dres}; 3 warm-up + 2 timed launches each) twice under `rocprofv3 --pmc`, once per counter (FETCH_SIZE and WRITE_SIZE
do not fit one pass — MI355X_MICROARCH.md, HBM section), and pairs the dispatches with kbench's printed lines by order.
Corrections as in bench.py: both counters are KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, so it is
doubled; WRITE_SIZE is exact (profiles/r01b_pmc_kbench_cfg2_raw.json)."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^(dp_gn_relu_\S+(?: \+\w+)? \S+)\s+([\d.]+) ms\s+(\d+) B\s+([\d.]+) GB/s")


def one_pass(counter):
    exe = os.path.join(ROOT, "tools", "kbench")
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    d = tempfile.mkdtemp(prefix="gn_pmc_", dir="/tmp")
    try:
        p = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "kb", "--",
                            exe, "64", "32", "224", "2", "dp_gn_relu"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                           timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, check=True)
        lines = [m.groups() for m in (LINE.match(l.strip()) for l in p.stdout.splitlines()) if m]
        rows = []
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if "k_gn_relu" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
        rows.sort()
        return lines, rows
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        lines, rows = one_pass(counter)
        assert lines and len(rows) % len(lines) == 0, (len(lines), len(rows))
        per = len(rows) // len(lines)
        for i, (name, ms, algo, gbs) in enumerate(lines):
            chunk = rows[i * per:(i + 1) * per]
            kernels = sorted(set(re.sub(r"\(.*", "", k.replace("void (anonymous namespace)::", "")) for _, k, _ in chunk))
            e = out.setdefault(name, dict(entry=name, kernel=kernels, algorithmic_bytes=int(algo)))
            e[counter + "_KiB"] = round(sum(v for _, _, v in chunk[-2:]) / 2.0, 1)      # the 2 timed launches
            e["ms_under_rocprof_" + counter] = float(ms)
    table = []
    for e in out.values():
        e["traffic_bytes"] = int(round(1024 * (e["WRITE_SIZE_KiB"] + 2 * e["FETCH_SIZE_KiB"])))
        e["traffic_over_algorithmic"] = round(e["traffic_bytes"] / e["algorithmic_bytes"], 3)
        table.append(e)
    doc = dict(what="rocprofv3 --pmc (separate passes) over tools/kbench 64 32 224 2 dp_gn_relu: 256 samples per launch; "
                    "traffic = WRITE_SIZE + 2 x FETCH_SIZE (KiB -> B; gfx950 FETCH_SIZE correction)", entries=table)
    text = json.dumps(doc, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
    for e in table:
        print("%-36s %-40s algo %11d B  traffic %11d B  x%.3f" % (e["entry"], ",".join(e["kernel"])[:40], e["algorithmic_bytes"],
                                                                  e["traffic_bytes"], e["traffic_over_algorithmic"]))


if __name__ == "__main__":
    main()
