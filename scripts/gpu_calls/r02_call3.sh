#!/bin/bash
# Round 2, GPU call 3: determinism probe, GPU suite, kbench (stem, gn defaults), bench + rocprofv3 profile.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
( timeout 400 python scripts/determinism_probe.py ) > $O/determinism.jsonl 2> $O/determinism.err; echo "determinism rc=$?" | tee -a $O/rc.txt
( time timeout 900 python -m pytest tests -m gpu -q -rf -s --durations=8 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 stem ) > $O/kbench_stem.txt 2>&1; echo "kbench stem rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 dp_gn_relu ) > $O/kbench_gn_default.txt 2>&1; echo "kbench gn rc=$?" | tee -a $O/rc.txt
( time timeout 400 python bench.py --no-cpu-baseline --no-pmc ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; cat $O/determinism.jsonl | cut -c1-400; tail -12 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -2 $O/kbench_stem.txt; cut -c1-500 $O/bench.json; head -12 $O/kernel_stats_timed.txt | cut -c1-150
