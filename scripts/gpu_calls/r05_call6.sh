#!/bin/bash
# Round 5, GPU call 6: the whole GPU suite (no -x) with the batch-gated fold + MFMA route table; micro-batch 512 / 1024 / 2048
# (tile-count quantisation: 896 workgroups for 512 slots at 512 samples becomes 3584 at 2048).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05f; mkdir -p $O
for mb in 512 1024 2048 512 2048; do
  ( timeout 600 python bench.py --micro-batch $mb --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline ) > $O/bench_mb$mb.json 2> $O/bench_mb$mb.err; echo "bench mb$mb rc=$?" | tee -a $O/rc.txt
  python - $O/bench_mb$mb.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; rc=d.get("roofline_conv") or {}; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), {k:(v["ms_per_step"], v["tflops"]) for k,v in rc.get("own_conv_kernels",{}).items()})
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -2 $O/bench_mb$mb.err | cut -c1-300
done
( timeout 1800 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -40 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -25 $O/pytest_gpu.log
