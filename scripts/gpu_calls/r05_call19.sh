#!/bin/bash
# Round 5, GPU call 19: GroupNorm-apply of the FOLD kernels as packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 + v_max_f32:
# 8 instead of 12 VALU per staging item) — bit-identity with dp_gn_relu_fwd on the hardware, kbench, and the headline step
# against the previous build (gpurun_in/*_prev = call 18's candidate) on one box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05s; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fold_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "conv1x1 or conv3x3 or folded_graph or fold or headline or micro_batch" 2>&1 | tail -8 ) > $O/pytest_fold.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -3 $O/pytest_fold.log
( DP_C1_VARIANTS=0 timeout 200 gpurun_in/kbench_prev 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_prev.txt 2>&1; echo "kbench prev rc=$?" | tee -a $O/rc.txt
( DP_C1_VARIANTS=0 timeout 200 tools/kbench 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_new.txt 2>&1; echo "kbench new rc=$?" | tee -a $O/rc.txt
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
    if k in b and k[3] == "fold": print("%5s->%5s @%2s %-5s prev %.4f new %.4f ms  (%.3fx)" % (k + (a[k], b[k], a[k] / b[k])))
PY
( timeout 100 tools/kbench 512 1 224 20 conv3x3 ) > $O/kbench_conv3x3.txt 2>&1
grep -E "k_conv3x3_mfma fold" $O/kbench_conv3x3.txt | awk 'NR % 2 == 0' | cut -c1-120
( timeout 100 tools/kbench 512 1 224 20 conv3s2 ) > $O/kbench_conv3s2.txt 2>&1
grep -E "fold" $O/kbench_conv3s2.txt | awk 'NR % 2 == 0' | cut -c1-120
cp dorpatch_amd/lib/libdorpatch_hip.so /tmp/new.so
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
run new
cp gpurun_in/libdorpatch_hip_prev.so dorpatch_amd/lib/libdorpatch_hip.so
run prev
cp /tmp/new.so dorpatch_amd/lib/libdorpatch_hip.so
run new_b
cp gpurun_in/libdorpatch_hip_prev.so dorpatch_amd/lib/libdorpatch_hip.so
run prev_b
cp /tmp/new.so dorpatch_amd/lib/libdorpatch_hip.so
