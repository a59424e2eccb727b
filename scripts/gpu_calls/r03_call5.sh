#!/bin/bash
# Round 3, GPU call 5: the tests that failed in call 4 (null-distribution end metric with its report, headline thresholds),
# the real-backbone end metric, kbench affine (fma, exact reciprocal, untouched tiles skip the occlusion tests) + pooling mode 5.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03e; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_end_metric_gpu.py tests/test_headline_parity_gpu.py tests/test_placement_gpu.py tests/test_kernels_gpu.py -m gpu -q -rfs -s --tb=short -p no:cacheprovider -k "plausible or headline or shipped or placement or affine or pool or gn_relu" 2>&1 | grep -v "mask size" | tail -150 ) > $O/pytest_subset.log 2>&1
timeout 200 tools/kbench 64 32 224 10 affine > $O/kbench_affine.txt 2>&1
timeout 200 tools/kbench 64 32 224 10 pool > $O/kbench_pool.txt 2>&1
tail -70 $O/pytest_subset.log; grep -v calib $O/kbench_affine.txt $O/kbench_pool.txt | grep -v "check: mode"; grep -c "0 differing" $O/kbench_pool.txt
