#!/bin/bash
# Round 2, GPU call 10: TunableOp probe for the GEMM route; the new dp_apply_fwd default in the real bench (HIP-event
# time + live PMC traffic).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
( timeout 400 python scripts/tunableop_probe.py --csv $O/tunableop_gfx950.csv ) > $O/tunableop.jsonl 2> $O/tunableop.err; echo "tunableop rc=$?" | tee -a $O/rc.txt
( time timeout 500 python bench.py --no-cpu-baseline --no-sweep --steps 10 --warmup 3 ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
( timeout 300 python -m pytest tests -m gpu -q -x -k "apply or size_properties or hot_loop" -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; cut -c1-220 $O/tunableop.jsonl | tail -40; tail -3 $O/tunableop.err; ls -la $O/*.csv; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02j/bench.json")); print(d["ms_per_step"], d["value"], d["roofline"])
PY
tail -3 $O/pytest_gpu.log
