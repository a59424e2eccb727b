#!/bin/bash
# Round 5, GPU call 7: the step's independent micro-batches on 2 / 4 HIP streams (overlap of one micro-batch's HBM-bound kernels
# with another's matrix-core kernels); the headline-parity test with the new routes.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05g; mkdir -p $O
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
run mb512_s1 --micro-batch 512 --streams 1
run mb512_s2 --micro-batch 512 --streams 2
run mb512_s4 --micro-batch 512 --streams 4
run mb1024_s2 --micro-batch 1024 --streams 2
run mb256_s4 --micro-batch 256 --streams 4
run mb512_s2_b --micro-batch 512 --streams 2
run mb2048_s1 --micro-batch 2048 --streams 1
( timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q -rs -p no:cacheprovider 2>&1 | tail -8 ) > $O/pytest_headline.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -5 $O/pytest_headline.log
