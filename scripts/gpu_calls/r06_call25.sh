#!/bin/bash
# Round 6, GPU call 25: the headline step with larger micro-batches (1024 rows on two streams / one stream, 2048 rows) against
# the product's 512 x 2 streams — interleaved on one box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06z; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$? $(python -c "import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1)" | tee -a $O/rc25.txt; }
for rep in 1 2; do
  run mb512_s2_$rep
  run mb1024_s2_$rep --micro-batch 1024 --streams 2
  run mb1024_s1_$rep --micro-batch 1024 --streams 1
  run mb2048_s1_$rep --micro-batch 2048 --streams 1
  run mb768_s2_$rep --micro-batch 768 --streams 2
done
