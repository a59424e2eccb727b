#!/bin/bash
# GPU round 6 (<= ~4.5 min of box time, ordered by importance; each leg writes its own file so a cut-off
# call still returns the finished legs): GPU parity suite, smoke, bench line, knob A/B, rocprofv3 kernel stats.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
( time timeout 200 python -m pytest tests -m gpu -q -rf --durations=5 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( time timeout 120 python bench.py --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
( timeout 110 python scripts/ab_sweep.py --micro-batches 256,512,1024 --modes auto,auto@cublas,miopen --steps 3 ) > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +8M -delete
( timeout 90 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; tail -6 $O/pytest_gpu.log; cut -c1-1500 $O/bench.json; tail -2 $O/bench.err; cut -c1-400 $O/ab.jsonl | head -8; tail -2 $O/smoke.log; head -24 $O/kernel_stats_timed.txt | cut -c1-160
