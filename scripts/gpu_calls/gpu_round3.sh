#!/bin/bash
# GPU round 3: full GPU suite, kbench + PMC passes, final bench (sweep + cpu baseline), rocprofv3 stats,
# 2-rank functional test (gloo, one GPU), micro-batch / config-3 sensitivity.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests -m gpu -q -rf --durations=5 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 120 tools/kbench ) > $O/kbench.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
( timeout 120 tools/kbench 1 64 384 ) > $O/kbench_384.txt 2>&1
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o kbench -- $R/tools/kbench 64 32 224 2 > $O/pmc_$ctr.log 2>&1; echo "pmc $ctr rc=$?" | tee -a $O/rc.txt
done
cd $R
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
( time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --same-device --batch 4 --samples 8 --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
( timeout 200 python bench.py --batch 4 --samples 16 --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_1rank_small.json 2>> $O/bench_2rank_gloo.err
( timeout 300 python bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --micro-batch 512 ) > $O/bench_mb512.json 2> $O/bench_mb512.err; echo "mb512 rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --micro-batch 128 ) > $O/bench_mb128.json 2> $O/bench_mb128.err; echo "mb128 rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --batch 1 --samples 64 --size 384 --patch-budget 0.015625 --micro-batch 64 --steps 3 --warmup 1 --no-cpu-baseline ) > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?" | tee -a $O/rc.txt
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; tail -6 $O/pytest_gpu.log; cat $O/kbench.txt | grep -E "mask_stats|project|struct"; cut -c1-330 $O/bench.json; tail -3 $O/bench.err
cut -c1-200 $O/bench_2rank_gloo.json $O/bench_1rank_small.json $O/bench_mb512.json $O/bench_mb128.json $O/bench_cfg3.json; tail -5 $O/bench_2rank_gloo.err
ls $O/pmc_FETCH_SIZE/* | head
