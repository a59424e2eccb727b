#!/bin/bash
# Round 5, GPU call 18: k_conv1x1_mfma epilogue with the next fragment's residual requested before this fragment's stores, and
# no zeroing of pixel-less staging items — the committed build against the candidate (gpurun_in/*_next) on one box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05r; mkdir -p $O
( DP_C1_VARIANTS=0 timeout 200 tools/kbench 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_cur.txt 2>&1; echo "kbench cur rc=$?" | tee -a $O/rc.txt
( DP_C1_VARIANTS=0 timeout 200 gpurun_in/kbench_next 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_next.txt 2>&1; echo "kbench next rc=$?" | tee -a $O/rc.txt
python - $O <<'PY'
import re, sys
def rd(p):
    t = {}
    for l in open(p):
        m = re.match(r"dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x.* variant\s+(\d+) (\w+)\s+([\d.]+) ms", l)
        if m: t[(m.group(1), m.group(2), m.group(3), m.group(5))] = float(m.group(6))
    return t
a, b = rd(sys.argv[1] + "/kbench_conv1x1_cur.txt"), rd(sys.argv[1] + "/kbench_conv1x1_next.txt")
for k in a:
    if k in b: print("%5s->%5s @%2s %-5s cur %.4f next %.4f ms  (%.3fx)" % (k + (a[k], b[k], a[k] / b[k])))
PY
cp dorpatch_amd/lib/libdorpatch_hip.so /tmp/cur.so
run() {  # name, args
  name=$1; shift
  ( timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
run cur
cp gpurun_in/libdorpatch_hip_next.so dorpatch_amd/lib/libdorpatch_hip.so
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fold_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "conv1x1 or folded_graph or fold or headline or micro_batch" 2>&1 | tail -8 ) > $O/pytest_next.log 2>&1; echo "pytest next rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -3 $O/pytest_next.log
run next
cp /tmp/cur.so dorpatch_amd/lib/libdorpatch_hip.so
run cur_b
cp gpurun_in/libdorpatch_hip_next.so dorpatch_amd/lib/libdorpatch_hip.so
run next_b
cp /tmp/cur.so dorpatch_amd/lib/libdorpatch_hip.so
