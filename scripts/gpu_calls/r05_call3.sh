#!/bin/bash
# Round 5, GPU call 3: calibration of the MFMA loop structure (k_mfma_probe: registers only / + LDS operand reads / + barrier /
# + LDS stores, at 1 and 2 workgroups per CU) and the 1x1 kernel with the LDS staging spread over the MFMA groups (variant 0)
# against one lump (variant 8), plain / folded GroupNorm / epilogue add, every shape.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05c; mkdir -p $O
( timeout 120 tools/kbench 512 1 224 5 mfma_probe ) > $O/kbench_mfma_probe.txt 2>&1; echo "probe rc=$?" | tee -a $O/rc.txt
cat $O/kbench_mfma_probe.txt
( DP_C1_VARIANTS=0,8 timeout 300 tools/kbench 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_spread.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
python - $O/kbench_conv1x1_spread.txt <<'PY'
import re, sys, collections
t = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x.* variant\s+(\d+) (\w+)\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m: t.setdefault((m.group(1), m.group(2), m.group(3)), {})[(m.group(5), int(m.group(4)))] = (float(m.group(6)), float(m.group(7)))
for k, v in t.items(): print("%5s->%5s @%2s " % k + "  ".join("%s/v%d %.3f (%5.1f)" % (a[0], a[1], b[0], b[1]) for a, b in v.items()))
PY
( timeout 300 tools/kbench 512 1 224 10 conv3x3 ) > $O/kbench_conv3x3.txt 2>&1; cat $O/kbench_conv3x3.txt
