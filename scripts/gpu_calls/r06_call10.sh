#!/bin/bash
# Round 6, GPU call 10: Winograd kernel, second form (A operands straight from global memory in the lanes' operand order; only V
# goes through LDS): parity, timing against the direct kernel, the headline step.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "winograd" -x 2>&1 | tail -5 | tee $O/pytest_wino.log
timeout 900 python scripts/conv3x3_vs_miopen.py --wino 512 > $O/conv3x3_wino_vs_direct.jsonl 2> $O/conv3x3_wino.err; echo "wino rc=$?" | tee -a $O/rc.txt
python - $O/conv3x3_wino_vs_direct.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["shape"], d["ms"], d["speedup_over_direct"])
PY
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
run wino_on_1 X=1 --steps 10 --warmup 2 --no-sweep
