#!/bin/bash
# Round 6, GPU call 1: where the step's time goes at the SMALL batches: one-stream kernel traces of
# configs[3] (1 x 64), the reference's default size (1 x 128) and configs[0] (8 x 4) on the round-5 tree, + the headline.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06a; mkdir -p $O
prof() {  # name, timed steps, args
  name=$1; shift; st=$1; shift
  ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -o bench -- python $R/bench.py "$@" --streams 1 --steps $st --warmup 3 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_$name.json 2> $R/$O/prof_$name.err ); echo "prof $name rc=$?" | tee -a $O/rc.txt
  python scripts/rocpd_stats.py $(ls $O/prof_$name/*kernel_trace.csv | head -1) --timed-steps $st --top 90 > $O/kernel_stats_timed_$name.txt 2> $O/kernel_stats_$name.err
  find $O/prof_$name -name "*.csv" -size +1M -delete
}
prof cfg3 10 --config 3
prof b1s128 10 --batch 1 --samples 128
prof cfg0 10 --config 0
head -70 $O/kernel_stats_timed_cfg3.txt | cut -c1-150
run() {  # name, args
  name=$1; shift
  ( timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("micro_batch"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-300
}
run cfg3 --config 3 --steps 20 --warmup 3
run b1s128 --batch 1 --samples 128 --steps 20 --warmup 3
run cfg0 --config 0 --steps 20 --warmup 3
run cfg2 --config 2 --steps 10 --warmup 3
run headline --steps 10 --warmup 2
