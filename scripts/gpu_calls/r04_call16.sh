#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04p; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -10 ) | tee $O/pytest_gpu_last_tree.log
