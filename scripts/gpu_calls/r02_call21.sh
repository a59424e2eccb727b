#!/bin/bash
# Round 2, GPU call 21: skip_satisfied is now opt-in (bench + product default: every sample completes forward and
# backward), compaction only when >= 20 % of a group's samples are skippable.  The new GPU test, the full default
# bench line of this tree (live PMC traffic, sweep, CPU baseline), and per-step times of the opt-in path at
# several satisfied fractions.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02u
mkdir -p $O
cd $R
( time timeout 300 python -m pytest tests/test_skip_satisfied_gpu.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_skip.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( time timeout 420 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench full rc=$?" | tee -a $O/rc.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 16 --warmup 4"
timeout 200 $B --skip-satisfied on > $O/bench_skip_on.json 2> $O/bench_skip_on.err; echo "bench skip=on rc=$?" | tee -a $O/rc.txt
for f in 0.3 0.6; do
  timeout 200 $B --satisfied $f > $O/bench_satisfied_$f.json 2> $O/bench_satisfied_$f.err; echo "bench satisfied=$f rc=$?" | tee -a $O/rc.txt
done
cat $O/rc.txt; tail -4 $O/pytest_skip.log; cut -c1-3000 $O/bench.json; tail -6 $O/bench.err
for f in $O/bench_s*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    print(d["value"], d["ms_per_step"], d["config"]["backward"])
except Exception as e:
    print("unreadable:", e)
PY
done
