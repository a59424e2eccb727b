#!/bin/bash
# Round 3, GPU call 6: kbench (affine with windows in registers, pooling defaults), rocprofv3 kernel traces of the headline
# step, the placement step and configs[2]; short benches on the same box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
timeout 200 tools/kbench 64 32 224 10 pool > $O/kbench_pool.txt 2>&1
cd /tmp
for tag in headline placement cfg2; do
  case $tag in headline) extra="";; placement) extra="--placement";; cfg2) extra="--config 2 --steps 5";; esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc $extra > $O/prof_bench_$tag.json 2> $O/prof_$tag.err
  ts=3; [ $tag = cfg2 ] && ts=5
  python $R/scripts/rocpd_stats.py $(ls $O/prof_$tag/*kernel_trace.csv | head -1) --timed-steps $ts --top 60 > $O/kernel_stats_timed_$tag.txt 2> $O/kernel_stats_$tag.err
  find $O/prof_$tag -name "*kernel_trace.csv" -size +8M -delete
done
cd $R
timeout 300 python bench.py --placement --steps 5 --warmup 2 --no-sweep --no-pmc --no-cpu-baseline > $O/bench_placement.json 2> $O/bench_placement.err
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline --deterministic off > $O/bench_cfg2_detoff.json 2> $O/bench_cfg2_detoff.err
grep -v calib $O/kbench_affine.txt $O/kbench_pool.txt | grep -v "check: mode"
for tag in headline placement cfg2; do head -3 $O/kernel_stats_timed_$tag.txt | cut -c1-160; grep "k_apply\|k_pad_maxpool\|k_gn_relu_bwd_big\|bwd_stream\|k_stem" $O/kernel_stats_timed_$tag.txt | cut -c1-170 | sort -u | head -12; done
for f in $O/bench_*.json $O/prof_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"]["deterministic"][:60])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
