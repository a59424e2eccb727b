#!/bin/bash
# Round 5, GPU call 17 (final evidence): the default bench line (sweep, CPU baseline, live PMC traffic, conv roofline), a
# rocprofv3 kernel trace of the default command, the placement extension, the whole attack.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05q; mkdir -p $O
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - $O/bench.json <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
print(d["value"], d["ms_per_step"], d.get("step_tflops"), d["roofline"], d.get("cpu_baseline"), {k: v for k, v in (d.get("roofline_conv") or {}).items() if k != "classes"}, d.get("collect_failure_sweep_ms"), d.get("value_with_sweep_amortised"))
PY
tail -4 $O/bench.err | cut -c1-200
( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench.json 2> $R/$O/prof.err ); echo "prof rc=$?" | tee -a $O/rc.txt
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 --top 80 > $O/kernel_stats_timed_headline.txt 2> $O/kernel_stats.err
cp $(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
find $O/prof -name "*.csv" -size +1M -delete
head -30 $O/kernel_stats_timed_headline.txt | cut -c1-170
( timeout 300 python bench.py --placement --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_placement.json 2> $O/bench_placement.err; echo "placement rc=$?" | tee -a $O/rc.txt
python - $O/bench_placement.json <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
print("placement", d["value"], d["ms_per_step"], d["roofline"])
PY
( time timeout 900 python bench.py --whole-attack --attack-modes retire ) > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err; echo "whole-attack rc=$?" | tee -a $O/rc.txt
cut -c1-1500 $O/bench_whole_attack.json | tail -3
tail -3 $O/bench_whole_attack.err | cut -c1-200
