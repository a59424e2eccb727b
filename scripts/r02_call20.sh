#!/bin/bash
# Round 2, GPU call 20: the selected-sample backward (skip_satisfied): GPU suite, same-box A/B at full activity,
# what-if runs at 50 / 75 / 90 % satisfied samples, smoke.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 8 --warmup 3"
for rep in 1 2; do
  for mode in on off; do
    timeout 400 $B --skip-satisfied $mode > $O/bench_skip_${mode}_$rep.json 2> $O/bench_skip_${mode}_$rep.err; echo "bench skip=$mode rep $rep rc=$?" | tee -a $O/rc.txt
  done
done
for f in 0.5 0.75 0.9 0.97; do
  timeout 400 $B --satisfied $f > $O/bench_satisfied_$f.json 2> $O/bench_satisfied_$f.err; echo "bench satisfied=$f rc=$?" | tee -a $O/rc.txt
done
timeout 400 $B --satisfied 0.75 --skip-satisfied off > $O/bench_satisfied_0.75_noskip.json 2> $O/bench_satisfied_0.75_noskip.err; echo "bench satisfied=0.75 noskip rc=$?" | tee -a $O/rc.txt
( timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest_gpu.log
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    print(d["value"], d["ms_per_step"], d["config"]["backward"], d["config"]["deterministic"])
except Exception as e:
    print("unreadable:", e)
PY
done
tail -3 $O/smoke.log
