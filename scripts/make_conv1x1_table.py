#!/usr/bin/env python
"""Derive dorpatch_amd/conv1x1_gfx950.json (route per GEMM batch, direction and shape) and
dorpatch_amd/tunableop_gfx950.csv (tuned GEMM solutions) from the committed probe outputs:

    python scripts/make_conv1x1_table.py profiles/r02l_tunableop_probe.jsonl:profiles/r02l_tunableop_raw_n512.csv [more pairs ...]

Each pair = the JSON lines printed by scripts/tunableop_probe.py (both routes timed in one process, GEMM with the default
and with the tuned solution) and the TunableOp CSV that run wrote.  Rules: a tuned solution is kept only if it measured
>= 1.02x the default (otherwise its CSV row is dropped and the default solution runs); plain column = faster of
(default GEMM, MIOpen); tuned column = faster of (kept GEMM time, MIOpen)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROW = re.compile(r"GemmStridedBatchedTunableOp_float_(\w+),(\w\w)_(\d+)_(\d+)_(\d+)_B_(\d+)_")


def main():
    plain, tuned, validators, gemm_rows, sources = {}, {}, None, [], []
    for pair in sys.argv[1:]:
        jl, csv_path = pair.split(":")
        sources.append(jl)
        runs = [json.loads(l) for l in open(jl) if l.startswith('{"dir"')]
        slow = set()
        for r in runs:
            n = str(r.get("N", 512))
            key = "%s:%d:%d:%d" % (r["dir"], r["C"], r["O"], r["HW"])
            keep = r["speedup"] >= 1.02
            g = min(r["gemm_tuned_ms"], r["gemm_default_ms"]) if keep else r["gemm_default_ms"]
            plain.setdefault(n, {})[key] = "gemm" if r["gemm_default_ms"] < r["miopen_ms"] else "miopen"
            tuned.setdefault(n, {})[key] = "gemm" if g < r["miopen_ms"] else "miopen"
            if not keep:   # (op, m, n, k, batch): forward = NN (HW, O, C), input gradient = NT (HW, C, O)
                slow.add(("NN", r["HW"], r["O"], r["C"], int(n)) if r["dir"] == "fwd" else ("NT", r["HW"], r["C"], r["O"], int(n)))
        seen = set()
        for line in open(csv_path).read().strip().split("\n"):
            m = ROW.match(line)
            if not m:
                if line.startswith("Validator,"):
                    validators = (validators or []) + ([line] if line not in (validators or []) else [])
                continue
            key = (m.group(1), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)))
            seen.add(key)
            if key not in slow and line not in gemm_rows:
                gemm_rows.append(line)
        assert slow <= seen, slow - seen
    doc = dict(arch="gfx950", key="GEMM batch -> direction:C:O:HW -> library route of the frozen stride-1 1x1 convolution",
               source="scripts/make_conv1x1_table.py over " + ", ".join(sources) +
                      " (scripts/tunableop_probe.py on one MI355X, ROCm 7.2, torch 2.10+rocm7.0, fp32)",
               plain=plain, tuned=tuned)
    with open(os.path.join(ROOT, "dorpatch_amd", "conv1x1_gfx950.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    with open(os.path.join(ROOT, "dorpatch_amd", "tunableop_gfx950.csv"), "w") as f:
        f.write("\n".join(validators + gemm_rows) + "\n")
    for n in sorted(plain, key=int):
        print("batch %s: %d shapes, plain gemm %d, tuned gemm %d" % (n, len(plain[n]), sum(v == "gemm" for v in plain[n].values()),
                                                                    sum(v == "gemm" for v in tuned[n].values())))
    print("%d tuned GEMM rows" % len(gemm_rows))


if __name__ == "__main__":
    main()
