#!/bin/bash
# Round 2, GPU call 27 (last of the round): the full GPU suite on the final tree.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02zz
mkdir -p $O
cd $R
( time timeout 125 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -6 $O/pytest_gpu.log
