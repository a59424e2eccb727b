#!/bin/bash
# Round 5, GPU call 10: k_conv3x3_flat (flat LDS image + masked taps) against k_conv3x3_mfma and MIOpen on the 224-input and
# 384-input planes, dp_conv3x3s2_bwd (the stride-2 input gradients) against MIOpen; parity tests; the headline step with the
# stride-2 input gradient on / off and the flat kernel on the 14 / 7 planes on / off.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05j; mkdir -p $O
( timeout 200 tools/kbench 512 1 224 20 conv3x3 ) > $O/kbench_conv3x3_rows_vs_flat.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
cut -c1-150 $O/kbench_conv3x3_rows_vs_flat.txt | awk 'NR % 2 == 0 || /^#/'
( DP_C3_SIDES=384 timeout 200 tools/kbench 64 1 224 20 conv3x3 ) > $O/kbench_conv3x3_flat_384_n64.txt 2>&1; echo "kbench384 rc=$?" | tee -a $O/rc.txt
cut -c1-150 $O/kbench_conv3x3_flat_384_n64.txt | awk 'NR % 2 == 0 || /^#/'
( timeout 200 tools/kbench 512 1 224 20 conv3s2bwd ) > $O/kbench_conv3s2bwd.txt 2>&1; echo "kbench s2bwd rc=$?" | tee -a $O/rc.txt
cut -c1-170 $O/kbench_conv3s2bwd.txt
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fold_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "conv3x3 or folded_graph" 2>&1 | tail -15 ) > $O/pytest_conv.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -6 $O/pytest_conv.log
( timeout 300 python scripts/conv3x3_vs_miopen.py --stride2 512 128 ) > $O/conv3x3s2_vs_miopen.jsonl 2> $O/conv3x3s2_vs_miopen.err; echo "vs_miopen s2 rc=$?" | tee -a $O/rc.txt
( timeout 300 python scripts/conv3x3_vs_miopen.py --flat 512 128 ) > $O/conv3x3_flat_vs_miopen.jsonl 2> $O/conv3x3_flat_vs_miopen.err; echo "vs_miopen flat rc=$?" | tee -a $O/rc.txt
( timeout 300 python scripts/conv3x3_vs_miopen.py --flat --384 64 128 ) > $O/conv3x3_flat384_vs_miopen.jsonl 2> $O/conv3x3_flat384_vs_miopen.err; echo "vs_miopen flat384 rc=$?" | tee -a $O/rc.txt
python - $O <<'PY'
import json, sys
for f in ("conv3x3s2_vs_miopen", "conv3x3_flat_vs_miopen", "conv3x3_flat384_vs_miopen"):
    for l in open("%s/%s.jsonl" % (sys.argv[1], f)):
        d = json.loads(l); print(d["shape"], d["ms"], d.get("speedup"), d.get("flat_equals_rows"), d.get("max_rel_diff_bwd_data", d.get("max_rel_diff_fwd")))
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
run s2bwd_on X=1
run s2bwd_off DORPATCH_CONV3X3S2_BWD=off
run s2bwd_on_flat X=1 --conv3x3-kernel flat
run s2bwd_on_s3 X=1 --streams 3
run s2bwd_on_flat_s3 X=1 --conv3x3-kernel flat --streams 3
