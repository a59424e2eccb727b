#!/bin/bash
# Round 2, GPU call 25: the opt-in selected-sample backward in a real DorPatch.generate run (1 image x 128 masks,
# both stages, 300 iterations each) vs the default; the README's main.py command with --skip_satisfied.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02y
mkdir -p $O
cd $R
timeout 400 python scripts/skip_real_run.py --iterations 300 > $O/skip_real_run.jsonl 2> $O/skip_real_run.err; echo "skip_real_run rc=$?" | tee -a $O/rc.txt
( cd /tmp && rm -rf mainrun && mkdir mainrun && cd mainrun && timeout 300 python $R/main.py --synthetic --num_images 1 --max_iterations 40 --targeted --quiet --skip_satisfied ) > $O/main_skip.log 2>&1; echo "main.py rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; cat $O/skip_real_run.jsonl; tail -5 $O/skip_real_run.err; tail -4 $O/main_skip.log
