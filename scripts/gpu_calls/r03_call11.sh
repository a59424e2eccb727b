#!/bin/bash
# Round 3, GPU call 18: rocprofv3 kernel trace of the placement step after the placement-kernel rewrite (the average
# duration of k_apply_affine_fwd must agree with the HIP-event figure in bench.py's roofline object).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_placement -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --placement > $O/prof_bench_placement.json 2> $O/prof_placement.err
python $R/scripts/rocpd_stats.py $(ls $O/prof_placement/*kernel_trace.csv | head -1) --timed-steps 3 --top 60 > $O/kernel_stats_timed_placement.txt 2> $O/kernel_stats_placement.err
find $O/prof_placement -name "*kernel_trace.csv" -size +8M -delete
cd $R
head -3 $O/kernel_stats_timed_placement.txt | cut -c1-160; grep "k_apply" $O/kernel_stats_timed_placement.txt | cut -c1-170
python - $O/prof_bench_placement.json <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(d["value"], d["ms_per_step"], d["roofline"])
PY
