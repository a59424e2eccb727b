#!/bin/bash
# Round 2, GPU call 9: apply_fwd samples-per-workgroup variants, GroupNorm forward V = 9 / 18 at 384x384 (configs[2]).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
( timeout 100 tools/kbench 64 32 224 20 apply_fwd ) > $O/kbench_apply.txt 2>&1; echo "kbench apply rc=$?" | tee -a $O/rc.txt
( timeout 300 python -m pytest tests -m gpu -q -x -k "gn or resnet or kernels" -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 200 python bench.py --config 2 --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --deterministic off ) > $O/bench_cfg2_off.json 2> $O/bench_cfg2_off.err; echo "bench cfg2 rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 170 rocprofv3 --kernel-trace --output-format csv -d $O/prof_cfg2 -o bench -- python $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --deterministic off > $O/prof_cfg2.json 2> $O/prof_cfg2.err; echo "rocprof cfg2 rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof_cfg2/*kernel_trace.csv | head -1) --timed-steps 5 > $O/kernel_stats_cfg2.txt 2>> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +6M -delete
cat $O/rc.txt; grep -E "apply_fwd" $O/kbench_apply.txt; tail -3 $O/pytest_gpu.log; cut -c1-200 $O/bench_cfg2_off.json; head -12 $O/kernel_stats_cfg2.txt | cut -c1-150
