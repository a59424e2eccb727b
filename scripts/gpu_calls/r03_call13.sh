#!/bin/bash
# Round 3, GPU call 20: the placement suite with the full-size launch test (64 x 32 x 224^2: walk invariance, adjoint
# identity in fp64, reproducibility).
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03u2; mkdir -p $O
( timeout 600 python -m pytest tests/test_placement_gpu.py -m gpu -q -x -s -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -12 ) > $O/pytest_placement.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
cat $O/pytest_placement.log
