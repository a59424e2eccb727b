#!/bin/bash
# GPU round 5 (budget-bounded, ~9 min): GPU parity suite, kernel micro-bench, the bench line, rocprofv3 kernel stats.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
( time timeout 330 python -m pytest tests -m gpu -q -rf --durations=8 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 60 tools/kbench ) > $O/kbench.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
( time timeout 200 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; tail -15 $O/pytest_gpu.log; grep -E "gn_relu|maxpool|stem|apply" $O/kbench.txt; cut -c1-900 $O/bench.json; tail -3 $O/bench.err; head -30 $O/kernel_stats_timed.txt
