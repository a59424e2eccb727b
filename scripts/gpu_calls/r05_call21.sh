#!/bin/bash
# Round 5, GPU call 21 (last): the N > 1 path on the last tree as far as one GPU allows — RCCL executed in a world-size-1 group
# (every collective of the step through dorpatch_amd.dist), and 2 ranks on the one GPU over gloo (the plain --gpus 2 command).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05u; mkdir -p $O
F="--steps 4 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline"
( timeout 200 python bench.py --gpus 1 --backend nccl --force-pg $F ) > $O/bench_world1_nccl.json 2> $O/bench_world1_nccl.err; echo "world1 nccl rc=$?" | tee -a $O/rc.txt
( timeout 200 python bench.py --gpus 1 --backend nccl --force-pg --comm-only ) > $O/bench_comm_only_world1.json 2> $O/bench_comm_only_world1.err; echo "comm-only rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --gpus 2 --same-device --backend gloo $F ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank gloo rc=$?" | tee -a $O/rc.txt
for f in bench_world1_nccl bench_comm_only_world1 bench_2rank_gloo; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d.get("metric"), d["value"], d.get("ms_per_step"), d.get("n_gpus"), str(d["config"].get("parallelism"))[:150])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -2 $O/bench_2rank_gloo.err | cut -c1-200
