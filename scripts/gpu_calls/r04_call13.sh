#!/bin/bash
# Round 4, GPU call 13 (last tree): full GPU suite, smoke, the default bench line.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04m; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -14 ) > $O/pytest_gpu_final_rs.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -8 $O/pytest_gpu_final_rs.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -1 $O/smoke.log
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["config"]["conv3x3"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"], d["roofline_project_update"]["frac"], d["roofline_project_update"]["traffic"], d["cpu_baseline"]["value"], d["collect_failure_sweep_ms"])
PY
