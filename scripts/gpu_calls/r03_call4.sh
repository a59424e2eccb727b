#!/bin/bash
# Round 3, GPU call 4: suite after the head convolution joined the determinism policy; det_trace again; kbench of the
# float4-staged affine kernels and the pooling launch modes (with byte-equality checks).
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03d; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -rfs --tb=short -p no:cacheprovider -k "not through_resnetv2" 2>&1 | tail -150 ) > $O/pytest_gpu.log 2>&1
timeout 300 python scripts/det_trace.py --n 8 128 > $O/det_trace.jsonl 2> $O/det_trace.err
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
timeout 200 tools/kbench 64 32 224 10 pool > $O/kbench_pool.txt 2>&1
grep -v "^\.\|^$" $O/pytest_gpu.log | tail -60; cat $O/det_trace.jsonl; grep -v calib $O/kbench_affine.txt $O/kbench_pool.txt
