#!/bin/bash
# Round 6, GPU call 6: which of the two tile rules costs the two-stream headline step its 1.5 % (call 5)?  Same box, interleaved:
# (a) round-5 tiles everywhere, (b) 3x3 rule only, (c) 1x1 rule only, (d) both rules.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06f; mkdir -p $O
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" >> $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("micro_batch"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in 1 2 3; do
run a_r5tiles_$rep DORPATCH_BENCH_DEBUG_SET=5=16,6=16 --steps 10 --warmup 2
run b_c3rule_$rep DORPATCH_BENCH_DEBUG_SET=5=16 --steps 10 --warmup 2
run c_c1rule_$rep DORPATCH_BENCH_DEBUG_SET=6=16 --steps 10 --warmup 2
run d_rules_$rep X=1 --steps 10 --warmup 2
done
