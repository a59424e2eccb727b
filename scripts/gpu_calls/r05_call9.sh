#!/bin/bash
# Round 5, GPU call 9: kernel traces of the present tree (all own convolutions incl. the stride-2 3x3 forward): one stream
# (clean per-kernel durations) and the default two streams; stream-count / micro-batch scan (the r05g files were lost with
# the container: re-recorded here).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05i; mkdir -p $O
for s in 1 2; do
  ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_s$s -o bench -- python $R/bench.py --streams $s --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_s$s.json 2> $R/$O/prof_s$s.err ); echo "prof s$s rc=$?" | tee -a $O/rc.txt
  python scripts/rocpd_stats.py $(ls $O/prof_s$s/*kernel_trace.csv | head -1) --timed-steps 3 --top 80 > $O/kernel_stats_timed_headline_streams$s.txt 2> $O/kernel_stats_s$s.err
  find $O/prof_s$s -name "*.csv" -size +1M -delete
done
head -60 $O/kernel_stats_timed_headline_streams1.txt | cut -c1-170
run() {  # name, args
  name=$1; shift
  ( timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("micro_batch"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-300
}
run mb512_s3 --micro-batch 512 --streams 3
run mb512_s4 --micro-batch 512 --streams 4
run mb256_s4 --micro-batch 256 --streams 4
run mb1024_s2 --micro-batch 1024 --streams 2
