#!/bin/bash
# Round 5, GPU call 22: the end metric at 224 x 224 on the extended fixture (6 images x 4 unmodified-reference runs).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05v; mkdir -p $O
( timeout 175 python -m pytest tests/test_end_metric_gpu.py -m gpu -q -s -p no:cacheprovider -k "224" 2>&1 | tail -30 ) > $O/pytest_end_metric_224.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -14 $O/pytest_end_metric_224.log | cut -c1-400
