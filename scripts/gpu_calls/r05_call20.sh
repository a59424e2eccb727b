#!/bin/bash
# Round 5, GPU call 20: the whole GPU suite + smoke() on the LAST tree of the round.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05t; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | tail -40 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -25 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
tail -3 $O/smoke.log
