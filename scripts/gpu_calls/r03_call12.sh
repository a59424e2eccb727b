#!/bin/bash
# Round 3, GPU call 19: the driver's own line on the final tree — default `python bench.py` (live PMC traffic, CPU baseline).
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03t; mkdir -p $O
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cut -c1-2500 $O/bench.json; tail -5 $O/bench.err
