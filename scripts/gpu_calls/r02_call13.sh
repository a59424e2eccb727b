#!/bin/bash
# Round 2, GPU call 13 (final validation of the round: same legs as call 7): GPU suite (placement extension, measured determinism), the full default bench line (live PMC
# traffic + CPU baseline), deterministic off for comparison on the same box, rocprofv3 kernel trace, the other
# single-GPU configs, the 2-rank bench on one device.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q -rf -s --durations=5 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
( time timeout 500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --deterministic off ) > $O/bench_det_off.json 2> $O/bench_det_off.err; echo "bench det off rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_auto.json 2> $O/bench_auto.err; echo "bench auto rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 --top 70 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
cp $O/prof/bench_kernel_stats.csv $O/rocprofv3_kernel_stats_wholeprocess.csv 2>/dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete
for c in 2 3 0; do ( timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?" | tee -a $O/rc.txt; done
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-pmc --samples 16 ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; tail -6 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-2600 $O/bench.json; tail -5 $O/bench.err
for f in bench_det_off bench_auto bench_cfg2 bench_cfg3 bench_cfg0 bench_2rank_gloo; do echo $f; python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d["config"].get("deterministic"))
except Exception as e: print("ERR", e)
PY
done
head -14 $O/kernel_stats_timed.txt | cut -c1-150
