#!/bin/bash
# Round 2, GPU call 14: kernel trace of the collect_failure sweep ("the largest single item nobody has profiled").
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02n
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o sweep -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/bench_sweep.json 2> $O/bench_sweep.err; echo "rocprof sweep rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/sweep_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) > $O/sweep_kernel_stats.txt 2> $O/sweep_stats.err
find $O -name "*kernel_trace.csv" -delete
cat $O/rc.txt; cat $O/sweep_kernel_stats.txt; tail -3 $O/bench_sweep.err; cut -c1-200 $O/bench_sweep.json
