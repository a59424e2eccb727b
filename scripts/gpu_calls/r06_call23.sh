#!/bin/bash
# NOTE: records an experiment whose code was NOT kept (the kbench variants / switches it names are described in DESIGN.md section 9 and in the profiles it wrote); it does not run on the committed tree as is.
# Round 6, GPU call 23: host time per launch — raw stream handle + memoised _sum_ok against the previous host path
# (DORPATCH_AB_OLD_HOST=1, experiment switch), interleaved on one box: configs[0], configs[3], 1 x 128, and the headline.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06y; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1)" | tee -a $O/rc23.txt; }
for rep in 1 2 3; do
  DORPATCH_AB_OLD_HOST=1 run cfg0_old_$rep --config 0 --steps 200 --warmup 10
  run cfg0_new_$rep --config 0 --steps 200 --warmup 10
  DORPATCH_AB_OLD_HOST=1 run cfg3_old_$rep --config 3 --steps 100 --warmup 10
  run cfg3_new_$rep --config 3 --steps 100 --warmup 10
  DORPATCH_AB_OLD_HOST=1 run b1s128_old_$rep --batch 1 --samples 128 --steps 60 --warmup 10
  run b1s128_new_$rep --batch 1 --samples 128 --steps 60 --warmup 10
done
DORPATCH_AB_OLD_HOST=1 run cfg1_old --steps 5 --warmup 2
run cfg1_new --steps 5 --warmup 2
