#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04h; mkdir -p $O
( timeout 300 python scripts/conv3x3_sustained.py ) > $O/conv3x3_sustained.json 2> $O/err.txt; cat $O/conv3x3_sustained.json; tail -2 $O/err.txt
