#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04n; mkdir -p $O
( timeout 300 python scripts/conv3x3_vs_miopen.py 64 256 ) > $O/conv3x3_vs_miopen_n64_n256.jsonl 2> $O/conv3x3.err; cut -c1-300 $O/conv3x3_vs_miopen_n64_n256.jsonl
