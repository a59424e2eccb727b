#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04k; mkdir -p $O
( timeout 300 python scripts/conv3x3_vs_miopen.py 512 128 ) > $O/conv3x3_vs_miopen.jsonl 2> $O/conv3x3.err; cut -c1-330 $O/conv3x3_vs_miopen.jsonl
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv3x3" 2>&1 | tail -3 ) | tee $O/pytest_conv.log
