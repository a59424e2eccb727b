#!/bin/bash
# Round 2, GPU call 5: apply_fwd variants (channel-split, contiguous 28 KiB per WG per sample) with byte-equality check,
# pooling kernels after the index-math rewrite, 2-rank check with robust statistics, GPU suite, bench + rocprofv3.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
( timeout 100 tools/kbench 64 32 224 20 apply_fwd ) > $O/kbench_apply.txt 2>&1; echo "kbench apply rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 1 64 384 20 apply_fwd ) > $O/kbench_apply_384.txt 2>&1; echo "kbench apply 384 rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 maxpool ) > $O/kbench_maxpool.txt 2>&1; echo "kbench maxpool rc=$?" | tee -a $O/rc.txt
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 scripts/two_rank_check.py ) > $O/two_rank_check.json 2> $O/two_rank_check.err; echo "2rank rc=$?" | tee -a $O/rc.txt
( time timeout 900 python -m pytest tests -m gpu -q -rf -s --durations=5 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
cd $R
python scripts/rocpd_stats.py $(ls $O/prof/*kernel_trace.csv | head -1) --timed-steps 3 --top 60 > $O/kernel_stats_timed.txt 2> $O/kernel_stats.err
find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/rc.txt; grep -E "apply_fwd" $O/kbench_apply.txt; grep -E "apply_fwd" $O/kbench_apply_384.txt; tail -3 $O/kbench_maxpool.txt; cut -c1-900 $O/two_rank_check.json; tail -5 $O/pytest_gpu.log; cut -c1-260 $O/bench.json; head -4 $O/kernel_stats_timed.txt | cut -c1-150
