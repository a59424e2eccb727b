#!/bin/bash
# Round 4, GPU call 7 (final tree): full GPU suite, smoke, the default bench line, its rocprofv3 kernel trace, the other
# single-GPU configs, the placement bench, the world-1 RCCL bench.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04g; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -25 ) > $O/pytest_gpu_final_rs.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -9 $O/pytest_gpu_final_rs.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -1 $O/smoke.log
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cat $O/bench.json | head -c 2500; echo
( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline > $R/$O/prof_bench_headline.json 2> $R/$O/prof_headline.err )
python scripts/rocpd_stats.py $(ls $O/prof_headline/*kernel_trace.csv | head -1) --timed-steps 3 --top 60 > $O/kernel_stats_timed_headline.txt 2> $O/kernel_stats.err
cp $(ls $O/prof_headline/*kernel_stats.csv 2>/dev/null | head -1) $O/rocprofv3_kernel_stats_headline.csv 2>/dev/null
find $O/prof_headline -name "*.csv" -size +1M -delete
head -28 $O/kernel_stats_timed_headline.txt | cut -c1-175
for cfg in 2 3 0; do
  ( timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline ) > $O/bench_cfg$cfg.json 2> $O/bench_cfg$cfg.err; echo "cfg$cfg rc=$?" | tee -a $O/rc.txt
done
( timeout 300 python bench.py --placement --steps 5 --warmup 2 --no-sweep --no-pmc --no-cpu-baseline ) > $O/bench_placement.json 2> $O/bench_placement.err; echo "placement rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --gpus 1 --backend nccl --force-pg --steps 5 --warmup 2 --no-sweep --no-pmc --no-cpu-baseline ) > $O/bench_world1_nccl.json 2> $O/bench_world1_nccl.err; echo "world1 nccl rc=$?" | tee -a $O/rc.txt
for f in $O/bench_cfg*.json $O/bench_placement.json $O/bench_world1_nccl.json $O/prof_bench_headline.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"]["conv3x3"])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
