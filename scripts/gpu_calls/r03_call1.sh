#!/bin/bash
# Round 3, GPU call 1: sanity of the refactors (suite with -rs), RCCL world-size-1, kbench baselines for the affine
# kernels + cold 512-sample pooling, per-problem determinism at the reference's own problem size.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03a; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | tail -60 ) > $O/pytest_gpu.log 2>&1
timeout 300 python scripts/rccl_world1.py > $O/rccl_world1.json 2> $O/rccl_world1.err
timeout 400 python bench.py --gpus 1 --backend nccl --force-pg --steps 3 --warmup 2 --no-pmc --no-cpu-baseline > $O/bench_world1_nccl.json 2> $O/bench_world1_nccl.err
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
timeout 200 tools/kbench 64 32 224 10 pool > $O/kbench_pool.txt 2>&1
for det in auto off on; do
  timeout 300 python bench.py --batch 1 --samples 128 --steps 30 --warmup 6 --no-sweep --no-pmc --no-cpu-baseline --deterministic $det > $O/bench_b1s128_$det.json 2> $O/bench_b1s128_$det.err
done
tail -5 $O/pytest_gpu.log; cat $O/rccl_world1.json; cat $O/kbench_affine.txt $O/kbench_pool.txt | grep -v calib
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["deterministic"], d["config"]["conv1x1"]["gemm_solutions"])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
