#!/bin/bash
# Round 6, GPU call 24: the last tree (host-path change + the stream-handle test): whole GPU suite, smoke, the default bench line.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
( timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench default rc=$?" | tee -a $O/rc.txt
tail -c 600 $O/bench_default.err
run() { name=$1; shift; ( timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$? $(python -c "import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1)" | tee -a $O/rc.txt; }
run cfg3 --config 3 --steps 100 --warmup 10
run b1s128 --batch 1 --samples 128 --steps 60 --warmup 10
run cfg0 --config 0 --steps 200 --warmup 10
run cfg2 --config 2 --steps 10 --warmup 3
