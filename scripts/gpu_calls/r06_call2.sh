#!/bin/bash
# Round 6, GPU call 2: dp_conv1x1_fwd with 448 / 256 / 128 / 64-pixel tiles (DP_DEBUG_CONV1X1_VARIANT bits 4-6) at
# N = 32 / 64 / 128 / 256 / 512, plain + fold + res, every 1x1 shape of ResNetV2-50 (both directions = both (C, O) orders);
# the parity tests of the new tiles.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv1x1" -x 2>&1 | tail -5 | tee $O/pytest_conv1x1.log
for n in 64 32 128 256 512; do
  DP_C1_VARIANTS=16,32,48,64,0 timeout 600 tools/kbench $n 1 224 20 conv1x1 > $O/kbench_conv1x1_tiles_n$n.txt 2>&1; echo "kbench n=$n rc=$?" | tee -a $O/rc.txt
done
head -40 $O/kbench_conv1x1_tiles_n64.txt
