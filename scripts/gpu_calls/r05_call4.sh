#!/bin/bash
# Round 5, GPU call 4: the MFMA loop probe with global loads (one burst per chunk vs one per MFMA group); the 1x1 kernel with
# BOTH the LDS stores and the global-load issue spread over the MFMA groups, branch-free (variant 0) vs staged in lumps
# (variant 8); headline A/B.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05d; mkdir -p $O
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
run_bench() {  # name, args...
  name=$1; shift
  ( timeout 400 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("gn_fold"), {k: v for k, v in d["config"].get("conv1x1", {}).items() if k in ("mode", "fwd", "bwd")})
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
run_bench r4routes --conv1x1 table --gn-fold off
run_bench mfma_fold --conv1x1 mfma --gn-fold on
run_bench mfma_nofold --conv1x1 mfma --gn-fold off
run_bench mfma_fold_2 --conv1x1 mfma --gn-fold on
