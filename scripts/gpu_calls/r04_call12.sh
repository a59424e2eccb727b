#!/bin/bash
# Round 4, GPU call 12: conv kernel after the selective register-pressure fix; the step with 7x7 (and 14x14) also routed.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04l; mkdir -p $O
( timeout 300 python scripts/conv3x3_vs_miopen.py 512 ) > $O/conv3x3_vs_miopen.jsonl 2> $O/conv3x3.err; cut -c1-330 $O/conv3x3_vs_miopen.jsonl
run() { tag=$1; shift; ( timeout 400 env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-sweep --no-update-roofline ) > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["config"]["conv3x3"], d["config"]["backward"]["step_ms"])
PY
}
run table DORPATCH_X=0
run plus7 DORPATCH_CONV3X3_ALSO=fwd:512:7,bwd:512:7
run plus7_14 DORPATCH_CONV3X3_ALSO=fwd:512:7,bwd:512:7,fwd:256:14,bwd:256:14
run off DORPATCH_CONV3X3=off
