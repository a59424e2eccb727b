#!/bin/bash
# Round 2, GPU call 11: tuned GEMM solutions (TunableOp file + route column) vs library defaults, same box, alternating.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
( timeout 300 python -m pytest tests -m gpu -q -x -k "conv1x1 or tuned or bit_identical or parity" -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
for rep in 1 2; do
  ( DORPATCH_TUNABLEOP=0 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_default_$rep.json 2> $O/bench_default_$rep.err; echo "bench default $rep rc=$?" | tee -a $O/rc.txt
  ( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_tuned_$rep.json 2> $O/bench_tuned_$rep.err; echo "bench tuned $rep rc=$?" | tee -a $O/rc.txt
done
cat $O/rc.txt; tail -3 $O/pytest_gpu.log
for f in bench_default_1 bench_tuned_1 bench_default_2 bench_tuned_2; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; print(sys.argv[1][-22:], d["ms_per_step"], d["value"], c["deterministic"], c["conv1x1"].get("gemm_solutions"), c["conv1x1"]["fwd"], c["conv1x1"]["bwd"])
PY
done
