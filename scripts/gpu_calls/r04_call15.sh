#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04o; mkdir -p $O
( timeout 900 python -m pytest tests/test_headline_parity_gpu.py tests/test_skip_satisfied_gpu.py tests/test_backbone_parity_gpu.py tests/test_backbone_vs_hf_bit_gpu.py tests/test_end_metric_gpu.py -m gpu -q -rs -p no:cacheprovider -k "not null and not certified_asr_matches" 2>&1 | grep -v "mask size" | tail -8 ) | tee $O/pytest_routes.log
