#!/bin/bash
# Round 2, GPU call 25 (second version: untimed warm-up of both paths first): the opt-in selected-sample backward in a real DorPatch.generate run (1 image x 128 masks,
# both stages, 200 iterations each) vs the default; the README's main.py command with --skip_satisfied.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02y
mkdir -p $O
cd $R
timeout 230 python scripts/skip_real_run.py --iterations 200 --warm 40 > $O/skip_real_run.jsonl 2> $O/skip_real_run.err; echo "skip_real_run rc=$?" | tee -a $O/rc.txt

cat $O/rc.txt; cat $O/skip_real_run.jsonl; tail -5 $O/skip_real_run.err; 
