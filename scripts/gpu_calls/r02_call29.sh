#!/bin/bash
# Round 2, GPU call 29: the one GPU test added after the final suite run (phase trace log).
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r02tr
timeout 30 python -m pytest tests/test_attack_gpu.py -m gpu -q -x -p no:cacheprovider -k "phase_trace" 2>&1 | tail -3 | tee gpurun_out/r02tr/pytest_phase.log
