#!/bin/bash
# Round 6, GPU call 4: the step at the small configurations with the new pixel tiles — route tables as committed (fold
# gated at 256 rows) vs the folded graph at every batch (DORPATCH_GNFOLD_MIN_BATCH=1: every convolution on own kernels).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06d; mkdir -p $O
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("micro_batch"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in a b; do
run cfg3_tab_$rep X=1 --config 3 --steps 20 --warmup 3
run cfg3_fold_$rep DORPATCH_GNFOLD_MIN_BATCH=1 --config 3 --steps 20 --warmup 3
run cfg3_fold_c3on_$rep "DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on" --config 3 --steps 20 --warmup 3
run b1s128_tab_$rep X=1 --batch 1 --samples 128 --steps 20 --warmup 3
run b1s128_fold_$rep DORPATCH_GNFOLD_MIN_BATCH=1 --batch 1 --samples 128 --steps 20 --warmup 3
run b1s128_fold_c3on_$rep "DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on" --batch 1 --samples 128 --steps 20 --warmup 3
run cfg0_tab_$rep X=1 --config 0 --steps 20 --warmup 3
run cfg0_fold_$rep DORPATCH_GNFOLD_MIN_BATCH=1 --config 0 --steps 20 --warmup 3
run cfg0_fold_c3on_$rep "DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on" --config 0 --steps 20 --warmup 3
done
run cfg2_tab X=1 --config 2 --steps 10 --warmup 3
run cfg2_fold "DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on" --config 2 --steps 10 --warmup 3
run headline X=1 --steps 10 --warmup 2
run headline_b X=1 --steps 10 --warmup 2
# one-stream kernel trace of the folded config 3
( cd /tmp; DORPATCH_GNFOLD_MIN_BATCH=1 DORPATCH_CONV3X3=on timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg3 -o bench -- python $R/bench.py --config 3 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_cfg3.json 2> $R/$O/prof_cfg3.err ); echo "prof rc=$?" | tee -a $O/rc.txt
python scripts/rocpd_stats.py $(ls $O/prof_cfg3/*kernel_trace.csv | head -1) --timed-steps 10 --top 90 > $O/kernel_stats_timed_cfg3_fold.txt 2> $O/kernel_stats.err
find $O/prof_cfg3 -name "*.csv" -size +1M -delete
