#!/bin/bash
# Round 2, GPU call 2: the full GPU parity suite (new: real-backbone fp64 parity, end-metric parity, bit
# reproducibility), smoke, kernel micro-benchmarks (copy calibration variants, GroupNorm variant sweep, new stem
# dgrad), the default bench line (live PMC traffic + CPU baseline) and a rocprofv3 kernel trace of the step.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q -rf -s --durations=8 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 calib ) > $O/kbench_calib.txt 2>&1; echo "kbench calib rc=$?" | tee -a $O/rc.txt
( timeout 200 tools/kbench 64 32 224 20 gn_relu ) > $O/kbench_gn.txt 2>&1; echo "kbench gn rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 stem ) > $O/kbench_stem.txt 2>&1; echo "kbench stem rc=$?" | tee -a $O/rc.txt
( time timeout 400 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; tail -25 $O/pytest_gpu.log; tail -3 $O/smoke.log; cat $O/kbench_calib.txt; cat $O/kbench_stem.txt | tail -3; cut -c1-1800 $O/bench.json; tail -4 $O/bench.err; head -30 $O/kernel_stats_timed.txt | cut -c1-150
