#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of the product build): one line per kernel
whose (demangled) name contains the filter.   python scripts/resusage.py /tmp/resusage.txt k_conv3x3_flat"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in text.split("Function Name: ")[1:]:
    name = b.split("\n")[0].split(" [")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    if flt not in dem:
        continue

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    dem = dem.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-60s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (
        dem[:60], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
