#!/bin/bash
# Round 2, GPU call 20 (lean: ~20 GPU-minutes were left): the selected-sample backward (skip_satisfied) on real
# hardware — GPU suite, same-box A/B at full activity (skip on = explicit tape, skip off = autograd), one what-if run
# at 75 % satisfied samples, smoke, rocprofv3 kernel trace of the default bench.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
( time timeout 420 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 8 --warmup 3"
for mode in on off; do
  timeout 240 $B --skip-satisfied $mode > $O/bench_skip_${mode}.json 2> $O/bench_skip_${mode}.err; echo "bench skip=$mode rc=$?" | tee -a $O/rc.txt
done
timeout 240 $B --satisfied 0.75 > $O/bench_satisfied_0.75.json 2> $O/bench_satisfied_0.75.err; echo "bench satisfied=0.75 rc=$?" | tee -a $O/rc.txt
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 --top 70 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
cp $O/prof/bench_kernel_stats.csv $O/rocprofv3_kernel_stats_wholeprocess.csv 2>/dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; tail -5 $O/pytest_gpu.log
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    print(d["value"], d["ms_per_step"], d["config"]["backward"], d["config"]["deterministic"], d["roofline"]["frac"])
except Exception as e:
    print("unreadable:", e)
PY
done
tail -3 $O/smoke.log; head -14 $O/kernel_stats_timed.txt | cut -c1-150
