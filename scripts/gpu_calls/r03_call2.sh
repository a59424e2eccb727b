#!/bin/bash
# Round 3, GPU call 2: full suite with failure list, determinism-probe protocols, kbench for the tiled affine kernels /
# pooling launch modes / 384^2 GroupNorm backward, bench --placement and configs[2].
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -rfs --tb=line -p no:cacheprovider 2>&1 | tail -80 ) > $O/pytest_gpu.log 2>&1
timeout 400 python scripts/det_probe2.py --n 8 16 > $O/det_probe2.jsonl 2> $O/det_probe2.err
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
timeout 200 tools/kbench 64 32 224 10 pool > $O/kbench_pool.txt 2>&1
timeout 200 tools/kbench 1 64 384 10 gn_relu > $O/kbench_gn384.txt 2>&1
timeout 400 python bench.py --placement --steps 3 --warmup 2 --no-sweep --no-pmc --no-cpu-baseline > $O/bench_placement.json 2> $O/bench_placement.err
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
grep -v "^\.\|^$" $O/pytest_gpu.log | tail -40; cat $O/det_probe2.jsonl; grep -v calib $O/kbench_affine.txt $O/kbench_pool.txt $O/kbench_gn384.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["config"]["deterministic"])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
