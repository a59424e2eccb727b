#!/bin/bash
# Round 5, GPU call 1: parity of the new 1x1 MFMA kernel (+ folded GroupNorm, epilogue add, launch variants) on the device;
# per-shape timing against the tuned library GEMMs / MIOpen with the launch variants; kernel trace of the round-4 default
# step (the evidence VERDICT r4 item 4a asks for) and the same step with every 1x1 convolution on the new kernel.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05a; mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rs -p no:cacheprovider -k "conv1x1" 2>&1 | tail -15 ) > $O/pytest_conv1x1.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -6 $O/pytest_conv1x1.log
( timeout 600 python scripts/conv1x1_vs_lib.py --variants 0,8,1,2,4 512 ) > $O/conv1x1_vs_lib_n512.jsonl 2> $O/conv1x1_vs_lib.err; echo "vs_lib rc=$?" | tee -a $O/rc.txt
python - $O/conv1x1_vs_lib_n512.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "dir" not in d: print(d); continue
    print(d["dir"], d["C"], d["O"], d["HW"], " ".join("%s=%.3f" % kv for kv in d["ms"].items()), "lib/mfma", d["lib_over_mfma"], "best", d["lib_over_best_variant"], "err %.1e" % d["max_rel_diff"])
PY
for mode in table mfma; do
  ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$mode -o bench -- python $R/bench.py --conv1x1 $mode --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline > $R/$O/prof_bench_$mode.json 2> $R/$O/prof_$mode.err ); echo "prof $mode rc=$?" | tee -a $O/rc.txt
  python scripts/rocpd_stats.py $(ls $O/prof_$mode/*kernel_trace.csv | head -1) --timed-steps 3 --top 70 > $O/kernel_stats_timed_$mode.txt 2> $O/kernel_stats_$mode.err
  find $O/prof_$mode -name "*.csv" -size +1M -delete
  head -30 $O/kernel_stats_timed_$mode.txt | cut -c1-170
done
for mode in table mfma table mfma; do
  ( timeout 400 python bench.py --conv1x1 $mode --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline ) > $O/bench_$mode.json 2> $O/bench_$mode.err; echo "bench $mode rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$mode.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("conv1x1"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
