#!/bin/bash
# Round 5, GPU call 14: the request-order fix of k_conv1x1_mfma<FOLD> / k_conv3x3s2_mfma (prologue and loop in one order, so the
# wait-count pass emits FIFO vmcnt values instead of draining the queue at every chunk): kbench of both builds, parity,
# the headline step with the previous and the new library on ONE box (gpurun_in/ holds the previous build).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05n; mkdir -p $O
SH="256:64:56,256:128:56,128:512:28,512:128:28,512:256:28,256:1024:14,1024:256:14,1024:512:14"
( DP_C1_SHAPES=$SH DP_C1_VARIANTS=0 timeout 200 gpurun_in/kbench_prev 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_prev.txt 2>&1; echo "kbench prev rc=$?" | tee -a $O/rc.txt
( DP_C1_SHAPES=$SH DP_C1_VARIANTS=0 timeout 200 tools/kbench 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_new.txt 2>&1; echo "kbench new rc=$?" | tee -a $O/rc.txt
python - $O <<'PY'
import re, sys
def rd(p):
    t = {}
    for l in open(p):
        m = re.match(r"dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x.* variant\s+(\d+) (\w+)\s+([\d.]+) ms", l)
        if m: t[(m.group(1), m.group(2), m.group(3), m.group(5))] = float(m.group(6))
    return t
a, b = rd(sys.argv[1] + "/kbench_conv1x1_prev.txt"), rd(sys.argv[1] + "/kbench_conv1x1_new.txt")
for k in a:
    if k in b: print("%5s->%5s @%2s %-5s prev %.4f new %.4f ms  (%.3fx)" % (k + (a[k], b[k], a[k] / b[k])))
PY
( timeout 100 gpurun_in/kbench_prev 512 1 224 20 conv3s2 ) > $O/kbench_conv3s2_prev.txt 2>&1
( timeout 100 tools/kbench 512 1 224 20 conv3s2 ) > $O/kbench_conv3s2_new.txt 2>&1
paste -d'|' <(cut -c1-95 $O/kbench_conv3s2_prev.txt) <(cut -c60-95 $O/kbench_conv3s2_new.txt) | awk 'NR % 2 == 0'
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fold_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "conv1x1 or stride2 or folded_graph or fold" 2>&1 | tail -15 ) > $O/pytest_conv.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -4 $O/pytest_conv.log
run() {  # name, args
  name=$1; shift
  ( timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-300
}
run new
cp dorpatch_amd/lib/libdorpatch_hip.so /tmp/new.so; cp gpurun_in/libdorpatch_hip_prev.so dorpatch_amd/lib/libdorpatch_hip.so
run prev
cp /tmp/new.so dorpatch_amd/lib/libdorpatch_hip.so
run new_b
cp gpurun_in/libdorpatch_hip_prev.so dorpatch_amd/lib/libdorpatch_hip.so
run prev_b
cp /tmp/new.so dorpatch_amd/lib/libdorpatch_hip.so
run new_s1 --streams 1
