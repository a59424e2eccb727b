#!/bin/bash
# Round 5, GPU call 11: dp_conv3x3s2_bwd in both forms (two column classes per workgroup with 8-byte interleaved stores /
# one class per workgroup) against MIOpen; parity; the headline step with it on / off on the same box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05k; mkdir -p $O
( timeout 200 tools/kbench 512 1 224 20 conv3s2bwd ) > $O/kbench_conv3s2bwd.txt 2>&1; echo "kbench s2bwd rc=$?" | tee -a $O/rc.txt
cut -c1-180 $O/kbench_conv3s2bwd.txt
( timeout 200 tools/kbench 128 1 224 20 conv3s2bwd ) > $O/kbench_conv3s2bwd_n128.txt 2>&1
cut -c1-180 $O/kbench_conv3s2bwd_n128.txt
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fold_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "stride2 or folded_graph" 2>&1 | tail -15 ) > $O/pytest_conv.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -4 $O/pytest_conv.log
( timeout 300 python scripts/conv3x3_vs_miopen.py --stride2 512 128 64 ) > $O/conv3x3s2_vs_miopen.jsonl 2> $O/conv3x3s2_vs_miopen.err; echo "vs_miopen s2 rc=$?" | tee -a $O/rc.txt
python - $O <<'PY'
import json, sys
for l in open("%s/conv3x3s2_vs_miopen.jsonl" % sys.argv[1]):
    d = json.loads(l); print(d["shape"], {k: v for k, v in d["ms"].items() if "bwd" in k}, d.get("speedup"), d.get("max_rel_diff_bwd_data"))
PY
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("conv3x3"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-300
}
run s2bwd_on DORPATCH_CONV3X3S2_BWD=on
run s2bwd_off DORPATCH_CONV3X3S2_BWD=off
run s2bwd_on_b DORPATCH_CONV3X3S2_BWD=on
run s2bwd_off_b DORPATCH_CONV3X3S2_BWD=off
run s2bwd_on_s1 DORPATCH_CONV3X3S2_BWD=on --streams 1
run s2bwd_off_s1 DORPATCH_CONV3X3S2_BWD=off --streams 1
