#!/bin/bash
# Round 2, GPU call 16: the per-batch route tables + tuned solutions on the other single-GPU configs (same-box A/B
# against DORPATCH_TUNABLEOP=0 + MIOpen-only routing = what ran there before), and the headline again.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
for c in 2 3 0; do
  ( DORPATCH_TUNABLEOP=0 timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 20 --warmup 5 --deterministic off --conv1x1 miopen ) > $O/bench_cfg${c}_miopen.json 2> /dev/null
  ( timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 20 --warmup 5 --deterministic off ) > $O/bench_cfg${c}_shipped_detoff.json 2> /dev/null
  ( timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 20 --warmup 5 ) > $O/bench_cfg${c}_shipped.json 2> /dev/null
done
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_cfg1_shipped.json 2> /dev/null
( timeout 400 python -m pytest tests -m gpu -q -x -k "conv1x1 or tuned or parity or 384 or attack" -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_gpu.log
for f in bench_cfg2_miopen bench_cfg2_shipped_detoff bench_cfg2_shipped bench_cfg3_miopen bench_cfg3_shipped_detoff bench_cfg3_shipped bench_cfg0_miopen bench_cfg0_shipped_detoff bench_cfg0_shipped bench_cfg1_shipped; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; print(sys.argv[1][-34:], d["ms_per_step"], d["value"], c["deterministic"][:12], c["conv1x1"]["fwd"], c["conv1x1"]["bwd"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
