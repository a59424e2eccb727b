#!/bin/bash
# Round 2, GPU call 24: HBM traffic of the GroupNorm+ReLU kernels from PMC counters (scripts/gn_pmc.py).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02x
mkdir -p $O
cd $R
timeout 400 python scripts/gn_pmc.py $O/gn_pmc.json > $O/gn_pmc.txt 2> $O/gn_pmc.err; echo "gn_pmc rc=$?" | tee -a $O/rc.txt
cat $O/gn_pmc.txt; tail -5 $O/gn_pmc.err
