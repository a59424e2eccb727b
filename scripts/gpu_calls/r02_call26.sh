#!/bin/bash
# Round 2, GPU call 26: the new reference-recorded fixtures (dual=True steps, untargeted run through the switch at
# iteration 500) on the GPU.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02z
mkdir -p $O
cd $R
( time timeout 200 python -m pytest tests/test_attack_gpu.py -m gpu -q -x -p no:cacheprovider -k "dual or untargeted or replays" ) > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -8 $O/pytest_new.log
