#!/bin/bash
# Round 2, GPU call 8: fused stem kernel after the slab change (kbench + same-box A/B against the unfused path).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
( timeout 100 tools/kbench 64 32 224 20 stem ) > $O/kbench_stem.txt 2>&1; echo "kbench stem rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --no-stem-split ) > $O/bench_unfused.json  # (flag since renamed: the unfused path is the default, --stem-split selects the fused kernel) 2> $O/bench_unfused.err; echo "bench unfused rc=$?" | tee -a $O/rc.txt
( timeout 300 python -m pytest tests -m gpu -q -x -k "stem or fused or parity" -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; tail -4 $O/kbench_stem.txt; tail -3 $O/pytest_gpu.log
for f in bench_fused bench_unfused; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1][-20:], d["ms_per_step"], d["value"])
PY
done
