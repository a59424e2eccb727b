#!/bin/bash
# Round 6, GPU call 12: with the Winograd 3x3 kernels in the step — stream count, the 1x1 tile rule against 448-pixel tiles
# everywhere, the sweep with 1024-row forwards; Winograd vs MIOpen on the 384-input planes.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06m; mkdir -p $O
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
run s2_$rep X=1 --steps 10 --warmup 2 --no-sweep
run s3_$rep X=1 --steps 10 --warmup 2 --no-sweep --streams 3
run s4_$rep X=1 --steps 10 --warmup 2 --no-sweep --streams 4
run s2_c1_448_$rep DORPATCH_BENCH_DEBUG_SET=5=16 --steps 10 --warmup 2 --no-sweep
run s1_$rep X=1 --steps 6 --warmup 2 --no-sweep --streams 1
done
run sweep_mb512 X=1 --steps 2 --warmup 1
run sweep_mb1024 X=1 --steps 2 --warmup 1 --micro-batch 1024
run sweep_s3 X=1 --steps 2 --warmup 1 --streams 3
timeout 600 python scripts/conv3x3_vs_miopen.py --wino --384 64 128 > $O/conv3x3_wino_384.jsonl 2> $O/conv3x3_wino_384.err; echo "wino384 rc=$?" | tee -a $O/rc.txt
python - $O/conv3x3_wino_384.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["shape"], d["ms"], d["speedup_over_direct"])
PY
