#!/bin/bash
# GPU round 2: kernel micro-bench, full GPU parity suite, smoke, bench A/B (fused GN on/off, MIOpen immediate mode),
# rocprofv3 kernel trace with CSV stats.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
( time timeout 120 tools/kbench ) > $O/kbench.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
( time timeout 900 python -m pytest tests -m gpu -q -rf --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
( time timeout 400 python bench.py --steps 3 --warmup 1 --no-sweep --no-cpu-baseline ) > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?" | tee -a $O/rc.txt
( time timeout 400 python bench.py --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-fused-gn ) > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
# keep only the small CSVs (the kernel trace itself can be large)
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
ls -laR $O/prof | head -20
cat $O/rc.txt; cat $O/kbench.txt; tail -15 $O/pytest_gpu.log; tail -3 $O/smoke.log; cat $O/bench_fused.json $O/bench_eager.json; tail -4 $O/bench_fused.err
