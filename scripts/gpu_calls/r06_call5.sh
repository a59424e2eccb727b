#!/bin/bash
# Round 6, GPU call 5: route-table inputs re-measured with the new pixel tiles (own kernels vs the libraries at 32 / 64 / 128 /
# 512 rows), the stride-2 forward's tiles, parity tests, and a same-box A/B of the headline step: round-5 tiles forced
# (DORPATCH_BENCH_DEBUG_SET=5=16,6=16: 448 pixels everywhere) vs the launchers' rule.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv1x1 or conv3x3" -x 2>&1 | tail -5 | tee $O/pytest_conv.log
timeout 900 python scripts/conv1x1_vs_lib.py 32 64 128 512 > $O/conv1x1_vs_lib.jsonl 2> $O/conv1x1_vs_lib.err; echo "c1 vs lib rc=$?" | tee -a $O/rc.txt
timeout 900 python scripts/conv3x3_vs_miopen.py 32 64 128 512 > $O/conv3x3_vs_miopen.jsonl 2> $O/conv3x3_vs_miopen.err; echo "c3 vs miopen rc=$?" | tee -a $O/rc.txt
timeout 900 python scripts/conv3x3_vs_miopen.py --stride2 32 64 128 512 > $O/conv3x3s2_vs_miopen.jsonl 2> $O/conv3x3s2_vs_miopen.err; echo "c3s2 vs miopen rc=$?" | tee -a $O/rc.txt
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("micro_batch"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in a b; do
run headline_r5tiles_$rep DORPATCH_BENCH_DEBUG_SET=5=16,6=16 --steps 10 --warmup 2
run headline_rule_$rep X=1 --steps 10 --warmup 2
run headline_r5tiles_s1_$rep DORPATCH_BENCH_DEBUG_SET=5=16,6=16 --steps 6 --warmup 2 --streams 1
run headline_rule_s1_$rep X=1 --steps 6 --warmup 2 --streams 1
done
run cfg3_fold "DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on" --config 3 --steps 20 --warmup 3
run cfg3_fold_r5tiles "DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on DORPATCH_BENCH_DEBUG_SET=5=16,6=16" --config 3 --steps 20 --warmup 3
run cfg3_tab X=1 --config 3 --steps 20 --warmup 3
