#!/usr/bin/env python
"""Derive dorpatch_amd/conv1x1_mfma_gfx950.json — which (GEMM batch, direction, C, O, HW) 1x1 convolutions run on the
hand-written fp32-MFMA kernel dp_conv1x1_fwd instead of the library route of conv1x1_gfx950.json — from the committed output
of scripts/conv1x1_vs_lib.py (one MI355X, same process, same tensors, tuned GEMM solutions active):

    python scripts/make_conv1x1_mfma_table.py profiles/r05e_conv1x1_vs_lib.jsonl [more ...]

Go / no-go per shape: mfma when best library time / mfma time >= 1.05 (box-to-box spread is +-3 %)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_GAIN = 1.05


def main():
    routes, sources = {}, []
    for path in sys.argv[1:]:
        sources.append(path)
        for line in open(path):
            d = json.loads(line)
            if "dir" not in d:
                continue
            ratio = min(d["ms"]["gemm"], d["ms"]["miopen"]) / d["ms"]["mfma"]
            col = routes.setdefault(str(d["N"]), {})
            if ratio >= MIN_GAIN:
                col["%s:%d:%d:%d" % (d["dir"], d["C"], d["O"], d["HW"])] = round(ratio, 3)
    doc = dict(arch="gfx950", key="GEMM batch -> direction:C:O:HW -> (best library ms / dp_conv1x1_fwd ms) of the shapes routed to "
                                  "the hand-written kernel (>= %.2f); a batch uses the column of the largest measured batch <= it" % MIN_GAIN,
               source="scripts/make_conv1x1_mfma_table.py over " + ", ".join(sources), routes=routes)
    with open(os.path.join(ROOT, "dorpatch_amd", "conv1x1_mfma_gfx950.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for n in sorted(routes, key=int):
        print("batch %s: %d shapes on the MFMA kernel" % (n, len(routes[n])))


if __name__ == "__main__":
    main()
