#!/bin/bash
# Round 4, GPU call 10: the default bench line on the last tree (PMC traffic for both roofline kernels), the 2-rank retire
# GPU tests.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04j; mkdir -p $O
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"], d["roofline_project_update"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
tail -12 $O/bench.err
( timeout 600 python -m pytest tests/test_attack_gpu.py tests/test_cabi.py -m gpu -q -rs -p no:cacheprovider -k "finished or retired" 2>&1 | tail -4 ) | tee $O/pytest_retire.log
