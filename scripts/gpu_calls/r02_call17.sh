#!/bin/bash
# Round 2, GPU call 17: does steering MIOpen's solver families change the step?  (same box, default first and last)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
run() { name=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?" | tee -a $O/rc.txt; }
run default A=1
run no_winograd MIOPEN_DEBUG_CONV_WINOGRAD=0
run no_igemm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0
run no_gemm MIOPEN_DEBUG_CONV_GEMM=0
run default2 A=1
for f in default no_winograd no_igemm no_gemm default2; do python - $O/bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["ms_per_step"], d["value"], d["config"]["deterministic"][:40])
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
