#!/bin/bash
# GPU round 4: validate the fused residual-add/GN, pad+maxpool, stem-dgrad kernels; CPU-baseline thread sweep;
# bench with kernel-precise apply timing; rocprofv3 stats.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
( time timeout 400 python -m pytest tests -m gpu -q -rf --durations=5 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 120 tools/kbench ) > $O/kbench.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
for t in 32 64; do ( time timeout 90 python bench.py --cpu-baseline-only --cpu-threads $t ) >> $O/cpu_threads.txt 2>&1; done
( time timeout 400 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
( timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --backend gloo --same-device --batch 4 --samples 8 --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
( timeout 200 python bench.py --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --no-fused-gn ) > $O/bench_eager.json 2> $O/bench_eager.err
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; tail -8 $O/pytest_gpu.log; grep -E "gn_relu|maxpool|stem|mask_stats" $O/kbench.txt; cat $O/cpu_threads.txt | grep -E "value|real"; cut -c1-700 $O/bench.json; tail -4 $O/bench.err; cut -c1-160 $O/bench_2rank_gloo.json $O/bench_eager.json
