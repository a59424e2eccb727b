#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python scripts/end_metric_debug.py > $O/end_metric_debug.txt 2> $O/end_metric_debug.err
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
cat $O/end_metric_debug.txt; tail -3 $O/end_metric_debug.err; grep -v calib $O/kbench_affine.txt
