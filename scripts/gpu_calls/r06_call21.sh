#!/bin/bash
# Round 6, GPU call 21 (the final tree, kernel sources split by family) — the whole GPU suite + smoke, the route measurements at
# N = 32 / 128 / 256 for the tables, the default bench line (sweep, PMC passes incl. the new 1x1 traffic classes, CPU baseline),
# one- and two-stream kernel traces of the default command, the other configurations.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
( timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench default rc=$?" | tee -a $O/rc.txt
tail -c 1500 $O/bench_default.json | head -c 1500; echo
for s in 1 2; do
  ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_s$s -o bench -- python $R/bench.py --streams $s --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_s$s.json 2> $R/$O/prof_s$s.err ); echo "prof s$s rc=$?" | tee -a $O/rc.txt
  python scripts/rocpd_stats.py $(ls $O/prof_s$s/*kernel_trace.csv | head -1) --timed-steps 3 --top 80 > $O/kernel_stats_timed_headline_streams$s.txt 2> $O/kernel_stats_s$s.err
  find $O/prof_s$s -name "*.csv" -size +1M -delete
done
head -40 $O/kernel_stats_timed_headline_streams1.txt | cut -c1-150
run() {  # name, args
  name=$1; shift
  ( timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" >> $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
run cfg3 --config 3 --steps 20 --warmup 3
run b1s128 --batch 1 --samples 128 --steps 20 --warmup 3
run cfg0 --config 0 --steps 20 --warmup 3
run cfg2 --config 2 --steps 10 --warmup 3

( timeout 900 python bench.py --whole-attack --attack-modes retire > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err ); echo "whole attack rc=$?" | tee -a $O/rc.txt
tail -c 900 $O/bench_whole_attack.json; echo
DP_C3_VARIANTS=1000,1 timeout 300 tools/kbench 512 1 224 10 conv3x3 > $O/kbench_conv3x3_wino_n512.txt 2>&1; tail -12 $O/kbench_conv3x3_wino_n512.txt
( timeout 600 python bench.py --gpus 2 --same-device --backend gloo --steps 5 --warmup 2 --no-sweep --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err ); echo "2-rank rc=$?" | tee -a $O/rc.txt
tail -c 400 $O/bench_2rank_gloo.json; echo
