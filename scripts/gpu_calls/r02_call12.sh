#!/bin/bash
# Round 2, GPU call 12: a longer TunableOp tuning pass (both routes timed in the same process), then the other configs
# with the shipped file (must not regress: the tuned route column only applies at the tuned batch size).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
( DORPATCH_TUNABLEOP=0 timeout 500 python scripts/tunableop_probe.py --csv $O/tunableop_gfx950.csv --max-ms 1500 --iters 60 ) > $O/tunableop.jsonl 2> $O/tunableop.err; echo "tunableop rc=$?" | tee -a $O/rc.txt
for c in 3 2; do
  ( DORPATCH_TUNABLEOP=0 timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --deterministic off ) > $O/bench_cfg${c}_default.json 2> /dev/null
  ( timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --deterministic off ) > $O/bench_cfg${c}_shipped.json 2> /dev/null
done
cat $O/rc.txt; cut -c1-260 $O/tunableop.jsonl | tail -32
for f in bench_cfg3_default bench_cfg3_shipped bench_cfg2_default bench_cfg2_shipped; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1][-26:], d["ms_per_step"], d["value"], d["config"]["conv1x1"])
PY
done
