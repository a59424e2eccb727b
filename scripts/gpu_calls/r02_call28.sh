#!/bin/bash
# Round 2, GPU call 28 (the last GPU-seconds of the round): DORPATCH_TRACE=1 — the roctx phase ranges on real hardware.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02tr
mkdir -p $O
cd $R
DORPATCH_TRACE=1 DORPATCH_TUNABLEOP=0 timeout 55 python bench.py --batch 1 --samples 16 --no-cpu-baseline --no-pmc --no-sweep --steps 2 --warmup 1 > $O/bench_trace.json 2> $O/bench_trace.err; echo "rc=$?" | tee $O/rc.txt
python -c "import json;d=json.loads(open('$O/bench_trace.json').readline());print('trace mode:',d['config']['trace'],d['ms_per_step'])"
tail -2 $O/bench_trace.err
