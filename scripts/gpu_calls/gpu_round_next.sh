#!/bin/bash
# First GPU call of the NEXT round (~12 min of box time): the measurements DESIGN.md §10 asks for before any
# NHWC work is written, then the knob A/B that round 6 could not finish, then the default bench + CPU baseline.
#   1. backbone_probe: frozen ResNetV2-50 fwd + input-grad bwd, NCHW vs channels_last (MIOpen told to use NHWC
#      kernels), eager GroupNorm in both so only the convolution layout differs          -> is NHWC worth building?
#   2. ab_sweep: conv1x1 route (auto / forced gemm / miopen) x BLAS library at micro-batch 512
#   3. bench.py (full line incl. cpu_baseline) and kbench
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/next
mkdir -p $O
cd $R
( PYTORCH_MIOPEN_SUGGEST_NHWC=1 timeout 300 python scripts/backbone_probe.py --mb 256 --find 0 --formats nchw nhwc --no-fused ) > $O/probe_layout.txt 2>&1; echo "probe rc=$?" | tee -a $O/rc.txt
# micro-batches small enough that one layer's activations (64 x 256 x 56^2 x 4 B = 205 MB) stay in the 256 MB
# Infinity Cache between producer and consumer kernels: do the HBM-bound GroupNorm kernels speed up, and does the
# host keep up (~450 launches per micro-batch)?  If the GPU wins but the host cannot: hipGraph the micro-batch.
( timeout 300 python scripts/ab_sweep.py --micro-batches 64,128 --modes auto --steps 3 ) > $O/ab_small_mb.jsonl 2> $O/ab_small_mb.err; echo "ab-small rc=$?" | tee -a $O/rc.txt
( timeout 240 python scripts/ab_sweep.py --micro-batches 512 --modes auto,gemm,miopen,auto@cublas --steps 3 ) > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?" | tee -a $O/rc.txt
# 2b. can the GEMM route go faster with PyTorch's TunableOp picking the hipBLASLt/rocBLAS solution per shape?
( timeout 60 python scripts/conv1x1_table.py ) > $O/conv1x1_default.jsonl 2> $O/c1.err; echo "table rc=$?" | tee -a $O/rc.txt
( PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=200 timeout 240 python scripts/conv1x1_table.py ) > $O/conv1x1_tunableop.jsonl 2> $O/c2.err; echo "table-tuned rc=$?" | tee -a $O/rc.txt
( time timeout 200 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
for c in 2 3; do ( timeout 150 python bench.py --config $c --no-cpu-baseline --no-sweep ) > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?" | tee -a $O/rc.txt; done
( timeout 60 tools/kbench ) > $O/kbench.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; cat $O/probe_layout.txt | tail -8; cut -c1-300 $O/ab.jsonl; cut -c1-600 $O/bench.json
