#!/bin/bash
# Round 5, GPU call 8: dp_conv3x3s2_fwd (the three stride-2 3x3 convolutions on the matrix cores, plain + GroupNorm fold):
# kbench timing, parity tests, per-shape timing against MIOpen, the headline step with the kernel on / off (2 streams default).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05h; mkdir -p $O
( timeout 120 tools/kbench 512 1 224 20 conv3s2 ) > $O/kbench_conv3s2.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
cat $O/kbench_conv3s2.txt | cut -c1-160
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fold_gpu.py tests/test_attack_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "stride2 or folded_graph or side_streams" 2>&1 | tail -15 ) > $O/pytest_s2.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -6 $O/pytest_s2.log
( timeout 300 python scripts/conv3x3_vs_miopen.py --stride2 512 128 ) > $O/conv3x3s2_vs_miopen.jsonl 2> $O/conv3x3s2_vs_miopen.err; echo "vs_miopen rc=$?" | tee -a $O/rc.txt
python - $O/conv3x3s2_vs_miopen.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["shape"], d["ms"], d["speedup"], d["max_rel_diff_fwd"], d["max_rel_diff_gn_fwd"])
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
run s2_on X=1
run s2_off DORPATCH_CONV3X3S2=off
run s2_on_s1 X=1 --streams 1
( timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q -rs -p no:cacheprovider 2>&1 | tail -8 ) > $O/pytest_headline.log 2>&1; echo "pytest headline rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -5 $O/pytest_headline.log
