#!/bin/bash
# Round 6, GPU call 3: k_conv3x3_flat with 448 / 128 / 64-pixel tiles vs k_conv3x3_mfma at N = 32 ... 512; dp_conv1x1_fwd with
# the launcher's measured tile rule; parity tests of both.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv1x1 or conv3x3" -x 2>&1 | tail -5 | tee $O/pytest_conv.log
for n in 64 32 128 256 512; do
  DP_C3_VARIANTS=1,16,48,64,0 timeout 600 tools/kbench $n 1 224 10 conv3x3 > $O/kbench_conv3x3_tiles_n$n.txt 2>&1; echo "kbench c3 n=$n rc=$?" | tee -a $O/rc.txt
  DP_C1_VARIANTS=0 timeout 600 tools/kbench $n 1 224 20 conv1x1 > $O/kbench_conv1x1_auto_n$n.txt 2>&1; echo "kbench c1 n=$n rc=$?" | tee -a $O/rc.txt
done
for n in 64 128; do
  DP_C3_SIDES=384 DP_C3_VARIANTS=16,48,64,0 timeout 600 tools/kbench $n 1 224 10 conv3x3 > $O/kbench_conv3x3_384_tiles_n$n.txt 2>&1; echo "kbench c3-384 n=$n rc=$?" | tee -a $O/rc.txt
done
grep -v "^#" $O/kbench_conv3x3_tiles_n64.txt | head -60
