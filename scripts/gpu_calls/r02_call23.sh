#!/bin/bash
# Round 2, GPU call 23: validation of the round's final tree (batch-128 route table + tuned solutions added, bool
# universe accepted by collect_failure, skip_satisfied opt-in): GPU suite, smoke, the reference-default problem size
# (1 image x 128 masks) after the batch-128 tables + its kernel trace (GPU-busy %), the headline configuration (short),
# the other single-GPU configs, 2 ranks on one device.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02w
mkdir -p $O
cd $R
( time timeout 420 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
Q="--no-cpu-baseline --no-pmc --no-sweep"
timeout 240 python bench.py --batch 1 --samples 128 $Q --steps 30 --warmup 6 > $O/bench_b1s128_after.json 2> $O/bench_b1s128_after.err; echo "bench b1s128 rc=$?" | tee -a $O/rc.txt
timeout 240 python bench.py --batch 1 --samples 128 $Q --steps 30 --warmup 6 --deterministic off > $O/bench_b1s128_after_detoff.json 2> $O/bench_b1s128_after_detoff.err; echo "bench b1s128 detoff rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b1s128 -- python $R/bench.py --batch 1 --samples 128 $Q --steps 20 --warmup 4 > $O/prof_b1s128.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 20 --top 40 > $O/kernel_stats_timed_b1s128.txt 2> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +8M -delete
timeout 300 python bench.py $Q --steps 10 --warmup 3 > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "bench cfg1 rc=$?" | tee -a $O/rc.txt
for c in 2 3 0; do timeout 200 python bench.py --config $c $Q --steps 20 --warmup 5 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?" | tee -a $O/rc.txt; done
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 $Q --samples 16 ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; head -3 $O/kernel_stats_timed_b1s128.txt
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["config"]["deterministic"], d["config"]["conv1x1"]["gemm_solutions"], d["roofline"]["frac"])
except Exception as e:
    print("unreadable:", e)
PY
done
