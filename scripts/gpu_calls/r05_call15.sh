#!/bin/bash
# Round 5, GPU call 15: the hit-compaction gather of dp_apply_affine_bwd (DP_DEBUG_AFFINE_GATHER = 2) in kbench + its
# bit-identity test; configs[2] (384 x 384, 1 x 64) with the round-5 routes on / off on one box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05o; mkdir -p $O
( timeout 200 tools/kbench 64 32 224 10 affine ) > $O/kbench_affine.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
cut -c1-150 $O/kbench_affine.txt
( timeout 600 python -m pytest tests/test_placement_gpu.py -m gpu -q -rs -x -p no:cacheprovider 2>&1 | tail -6 ) > $O/pytest_placement.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -3 $O/pytest_placement.log
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 600 python bench.py "$@" --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("conv3x3"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-200
}
run cfg2_r5 X=1 --config 2
run cfg2_r4routes "DORPATCH_CONV3X3S2_BWD=off DORPATCH_CONV3X3S2=off DORPATCH_STEM_CONV=off DORPATCH_CONV3X3_ALSO=" --config 2 --conv3x3-kernel rows
run cfg2_r5_b X=1 --config 2
run cfg0_r5 X=1 --config 0
run cfg3_r5 X=1 --config 3
