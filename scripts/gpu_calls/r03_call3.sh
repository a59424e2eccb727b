#!/bin/bash
# Round 3, GPU call 3: which call breaks run-to-run reproducibility under the per-problem policy; kbench of the reworked
# affine kernels, the pooling backward in row pairs, the large-group GroupNorm backward batch sizes.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03c; mkdir -p $O
timeout 400 python scripts/det_trace.py --n 8 32 128 > $O/det_trace.jsonl 2> $O/det_trace.err
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
timeout 200 tools/kbench 64 32 224 10 pool > $O/kbench_pool.txt 2>&1
timeout 200 tools/kbench 1 64 384 10 gn_relu > $O/kbench_gn384.txt 2>&1
cat $O/det_trace.jsonl; tail -3 $O/det_trace.err; grep -v calib $O/kbench_affine.txt $O/kbench_pool.txt $O/kbench_gn384.txt
