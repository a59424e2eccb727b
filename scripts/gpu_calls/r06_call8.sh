#!/bin/bash
# Round 6, GPU call 8: (1) the folded 1x1 launches on 256- / 128- / 64-pixel tiles inside the two-stream step (3 / 5 / 8 workgroups
# per CU: how much of the stand-alone gain survives next to the other stream's kernels?); (2) micro-batches of 256 / 128 rows
# with the new tiles (smaller activations stay in the 256 MB Infinity Cache between producer and consumer).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06h; mkdir -p $O
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 900 python bench.py "$@" --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" >> $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("collect_failure_sweep_ms"), d.get("value_with_sweep_amortised"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in 1 2; do
run base_$rep X=1 --steps 10 --warmup 2 --no-sweep
run fold256_$rep DORPATCH_BENCH_DEBUG_SET=5=128 --steps 10 --warmup 2 --no-sweep
run fold128_$rep DORPATCH_BENCH_DEBUG_SET=5=256 --steps 10 --warmup 2 --no-sweep
run mb256_$rep X=1 --steps 10 --warmup 2 --no-sweep --micro-batch 256
run mb128_$rep X=1 --steps 10 --warmup 2 --no-sweep --micro-batch 128
run mb256_s3_$rep X=1 --steps 10 --warmup 2 --no-sweep --micro-batch 256 --streams 3
done
run sweep_base X=1 --steps 2 --warmup 1
run sweep_fold256 DORPATCH_BENCH_DEBUG_SET=5=128 --steps 2 --warmup 1
run sweep_fold128 DORPATCH_BENCH_DEBUG_SET=5=256 --steps 2 --warmup 1
run sweep_fold64 DORPATCH_BENCH_DEBUG_SET=5=384 --steps 2 --warmup 1
