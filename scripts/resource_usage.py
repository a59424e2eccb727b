#!/usr/bin/env python
"""Per-kernel register / LDS / scratch / occupancy table of the gfx950 build, from hipcc's own
`-Rpass-analysis=kernel-resource-usage` remarks (needs no GPU).
    python scripts/resource_usage.py > profiles/rNN_kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dorpatch_amd import build  # noqa: E402


def main():
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-Wall")]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [build._hipcc()] + flags + ["-I", build.INCLUDE, "-Rpass-analysis=kernel-resource-usage", "-c", build.SRC,
                                          "-o", os.path.join(tmp, "x.o")]
        res = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
    if res.returncode != 0:
        sys.exit(res.stderr)
    names = re.findall(r"Function Name: (\S+)", res.stderr)
    for filt in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            names = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True,
                                   check=True).stdout.splitlines()
            break
        except (OSError, subprocess.CalledProcessError):
            continue
    blocks = re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]
    print("# %s" % " ".join(cmd[1:-4]))
    print("%-58s %5s %5s %5s %8s %10s %9s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "waves/SIMD", "LDS B/WG"))
    for name, b in zip(names, blocks):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\((?!anonymous).*", "", short).replace("void ", "")
        print("%-58s %5d %5d %5d %8d %10d %9d" % (short[:58], g("SGPRs"), g("VGPRs"), g("AGPRs"),
                                                  g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                                                  g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
