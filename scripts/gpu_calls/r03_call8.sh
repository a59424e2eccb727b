#!/bin/bash
# Round 3, GPU call 8: smoke() and the full default bench line (live PMC traffic, CPU baseline with the thread scan),
# bench at the reference's own size (per-problem determinism), configs 0 and 3.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03h; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
for det in auto on off; do timeout 300 python bench.py --batch 1 --samples 128 --steps 30 --warmup 6 --no-sweep --no-pmc --no-cpu-baseline --deterministic $det > $O/bench_b1s128_$det.json 2> $O/bench_b1s128_$det.err; done
for c in 0 3; do timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 20 --warmup 5 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
cat $O/rc.txt; tail -2 $O/smoke.log; cut -c1-3000 $O/bench.json; tail -6 $O/bench.err
for f in $O/bench_b1s128_*.json $O/bench_cfg*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["config"]["deterministic"][:70])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
