#!/bin/bash
# Round 5, GPU call 13: per-launch-class timing of every matrix-core convolution inside the step (one stream, events around
# each launch); micro-batch 512 vs 1024 on two streams, same box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05m; mkdir -p $O
run() {  # name, args
  name=$1; shift
  ( timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("micro_batch"))
    rc = d.get("roofline_conv")
    if rc and d["config"].get("streams") == 1:
        print(rc["kernel"], rc["launch_class"], rc["achieved"], rc["own_conv_share_of_step"])
        for k, v in rc["own_conv_kernels"].items(): print("   ", k, v)
        for c in rc["classes"]: print("   ", c)
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-300
}
run s1_classes --streams 1
run mb512_s2 --micro-batch 512 --streams 2 --no-conv-roofline
run mb1024_s2 --micro-batch 1024 --streams 2 --no-conv-roofline
run mb512_s2_b --micro-batch 512 --streams 2 --no-conv-roofline
run mb1024_s2_b --micro-batch 1024 --streams 2 --no-conv-roofline
run mb2048_s1 --micro-batch 2048 --streams 1 --no-conv-roofline
