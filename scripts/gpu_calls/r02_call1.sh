#!/bin/bash
# Round 2, GPU call 1 (measurement only, existing code): the single-GPU configs other than the headline
# (BASELINE configs[2] 384x384, configs[3] per-GPU share, configs[0]) with GPU-busy % from a rocprofv3 kernel
# trace, the NCHW-vs-NHWC convolution probe, the 2-rank product path on real kernels (gloo, one device), and
# Infinity-Cache-sized micro-batches.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
for c in 2 3 0; do
  ( timeout 170 python bench.py --config $c --no-cpu-baseline --no-sweep --steps 10 --warmup 3 ) > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?" | tee -a $O/rc.txt
done
cd /tmp
for c in 2 3; do
  timeout 170 rocprofv3 --kernel-trace --output-format csv -d $O/prof_cfg$c -o bench -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-sweep > $O/prof_cfg$c.json 2> $O/prof_cfg$c.err; echo "rocprof cfg$c rc=$?" | tee -a $O/rc.txt
  ( cd $R; python scripts/rocpd_stats.py $(ls $O/prof_cfg$c/*kernel_trace.csv | head -1) --timed-steps 5 > $O/kernel_stats_cfg$c.txt 2>> $O/kernel_stats.err )
  find $O/prof_cfg$c -name "*kernel_trace.csv" -size +6M -delete
done
cd $R
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --samples 16 ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
( PYTORCH_MIOPEN_SUGGEST_NHWC=1 timeout 300 python scripts/backbone_probe.py --mb 256 --find 0 --formats nchw nhwc --no-fused ) > $O/probe_layout.txt 2>&1; echo "probe rc=$?" | tee -a $O/rc.txt
( timeout 280 python scripts/ab_sweep.py --micro-batches 64,128 --modes auto --steps 3 ) > $O/ab_small_mb.jsonl 2> $O/ab_small_mb.err; echo "ab-small rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt
for c in 2 3 0; do cut -c1-400 $O/bench_cfg$c.json; done
head -3 $O/kernel_stats_cfg2.txt; head -3 $O/kernel_stats_cfg3.txt
cut -c1-300 $O/bench_2rank_gloo.json; tail -3 $O/bench_2rank_gloo.err
tail -4 $O/probe_layout.txt; cut -c1-200 $O/ab_small_mb.jsonl
