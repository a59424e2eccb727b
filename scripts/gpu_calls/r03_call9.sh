#!/bin/bash
# Round 3, GPU calls 9 / 17 / 25: validation of the round's final tree — full GPU suite with -rs (every skip named) on the
# regenerated fixtures, smoke(), short bench, 2 gloo ranks on one device.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03z; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q -rs -s --durations=8 -p no:cacheprovider 2>&1 | grep -v "mask size" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?" | tee -a $O/rc.txt
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-pmc --samples 16 ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
timeout 100 tools/kbench 64 32 224 10 pool 2>&1 | grep "512x64" > $O/kbench_pool_final.txt
cat $O/rc.txt; grep -E "passed|failed|SKIPPED|FAILED|certified|failures|cells|reaches" $O/pytest_gpu.log | tail -40; tail -2 $O/smoke.log
for f in bench_short bench_2rank_gloo; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("deterministic"))
except Exception as e: print("ERR", e)
PY
done
cat $O/kbench_pool_final.txt
