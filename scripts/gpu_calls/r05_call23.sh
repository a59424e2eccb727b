#!/bin/bash
# Round 5, GPU call 23: the plain multi-rank command WITHOUT --no-conv-roofline (what the driver's --gpus N run passes): the
# conv-roofline extra step must not run on a subset of the ranks (it holds the step's all-reduce).  2 ranks on the one GPU.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05w; mkdir -p $O
( timeout 100 python bench.py --gpus 2 --same-device --backend gloo --steps 3 --warmup 1 --no-sweep --no-pmc --no-update-roofline ) > $O/bench_2rank_default_flags.json 2> $O/bench_2rank_default_flags.err; echo "2rank rc=$?" | tee -a $O/rc.txt
python - $O/bench_2rank_default_flags.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(d["value"], d["ms_per_step"], d["n_gpus"], "roofline_conv" in d, "cpu_baseline" in d)
except Exception as e: print("unreadable", e)
PY
tail -3 $O/bench_2rank_default_flags.err | cut -c1-200
