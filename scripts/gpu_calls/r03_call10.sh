#!/bin/bash
# Round 3, GPU calls 10 / 11: placement forward after the occlusion-test rewrite (per-window uniform skip + bit-field mask; r03k) and with the sample walk (next footprint in flight; r03l):
# kbench, the placement GPU tests, the placement step by its own HIP events.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03y; mkdir -p $O
timeout 120 tools/kbench 64 32 224 20 affine > $O/kbench_affine.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
( timeout 600 python -m pytest tests/test_placement_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 ) > $O/pytest_placement.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
( timeout 300 python bench.py --placement --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_placement.json 2> $O/bench_placement.err; echo "bench rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; cat $O/kbench_affine.txt | tail -12; cat $O/pytest_placement.log; cut -c1-1500 $O/bench_placement.json
