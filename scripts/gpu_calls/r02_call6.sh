#!/bin/bash
# Round 2, GPU call 6: same-box A/B of the round-1 tree (git 7faf45e, exported beforehand with
#   mkdir .ab_r01 && git archive 7faf45e | tar -x -C .ab_r01 && (cd .ab_r01 && python -m dorpatch_amd.build)
# and deleted afterwards) against the current tree
# (alternating, so box-to-box clock differences cancel), MIOpen find mode, new GPU tests, micro-benchmarks.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
for rep in 1 2; do
  ( cd .ab_r01 && timeout 300 python bench.py --no-cpu-baseline --no-sweep --steps 10 --warmup 3 ) > $O/ab_r01_$rep.json 2> $O/ab_r01_$rep.err; echo "ab r01 $rep rc=$?" | tee -a $O/rc.txt
  ( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/ab_cur_$rep.json 2> $O/ab_cur_$rep.err; echo "ab cur $rep rc=$?" | tee -a $O/rc.txt
done
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --deterministic off ) > $O/ab_cur_nondet.json 2> $O/ab_cur_nondet.err; echo "ab cur nondet rc=$?" | tee -a $O/rc.txt
( time timeout 900 python -m pytest tests -m gpu -q -rf -s --durations=5 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 maxpool ) > $O/kbench_maxpool.txt 2>&1
( time timeout 700 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --find 1 ) > $O/bench_find1.json 2> $O/bench_find1.err; echo "bench find1 rc=$?" | tee -a $O/rc.txt
( du -sh ~/.config/miopen ~/.cache/miopen 2>/dev/null; find ~/.config/miopen -type f | head -20 ) > $O/miopen_userdb.txt 2>&1
mkdir -p $O/miopen_config && cp -r ~/.config/miopen/* $O/miopen_config/ 2>/dev/null; du -sh $O/miopen_config >> $O/miopen_userdb.txt
cat $O/rc.txt
for f in ab_r01_1 ab_cur_1 ab_r01_2 ab_cur_2 ab_cur_nondet bench_find1; do echo $f; cut -c1-150 $O/$f.json; done
tail -3 $O/bench_find1.err; tail -5 $O/pytest_gpu.log; tail -3 $O/kbench_maxpool.txt; cat $O/miopen_userdb.txt | head
