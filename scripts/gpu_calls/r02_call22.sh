#!/bin/bash
# Round 2, GPU call 22: the reference's DEFAULT problem size — one image x sampling_size = 128 masks (attack.py:53,
# main.py --batch_size 1; BASELINE configs[4] streams such images) — runs its 1x1 GEMM route at batch 128, for which
# no route table / tuned solutions exist yet.  Baseline bench at that size, then the route + TunableOp probe for it.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02v
mkdir -p $O
cd $R
B="python bench.py --batch 1 --samples 128 --no-cpu-baseline --no-pmc --no-sweep --steps 30 --warmup 6"
timeout 240 $B > $O/bench_b1s128_before.json 2> $O/bench_b1s128_before.err; echo "bench before rc=$?" | tee -a $O/rc.txt
timeout 240 $B --conv1x1 miopen > $O/bench_b1s128_miopen.json 2> $O/bench_b1s128_miopen.err; echo "bench miopen rc=$?" | tee -a $O/rc.txt
( DORPATCH_TUNABLEOP=0 timeout 400 python scripts/tunableop_probe.py --n 128 --size 224 --csv $O/tunableop_raw_n128_224.csv --max-ms 600 --iters 30 ) > $O/tunableop_probe_n128_224.jsonl 2> $O/tunableop_n128_224.err; echo "probe n=128 rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; tail -1 $O/tunableop_probe_n128_224.jsonl | cut -c1-200
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    print(d["value"], d["ms_per_step"], d["config"]["deterministic"], d["config"]["conv1x1"])
except Exception as e:
    print("unreadable:", e)
PY
done
