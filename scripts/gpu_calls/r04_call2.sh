#!/bin/bash
# Round 4, GPU call 2: full GPU suite; go/no-go of the MFMA 3x3 convolution against MIOpen; project_update after the
# prefetch reorder; whole attack with ladder-sized micro-batches and shape-stable sweeps.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04b; mkdir -p $O
( timeout 200 python scripts/conv3x3_vs_miopen.py 512 ) > $O/conv3x3_vs_miopen_n512.json 2> $O/conv3x3.err; echo "conv3x3 rc=$?" | tee -a $O/rc.txt
( timeout 100 python scripts/conv3x3_vs_miopen.py 128 ) > $O/conv3x3_vs_miopen_n128.json 2>> $O/conv3x3.err
cat $O/conv3x3_vs_miopen_n512.json $O/conv3x3_vs_miopen_n128.json; tail -3 $O/conv3x3.err
( timeout 100 tools/kbench 512 1 224 10 "conv3x3" ) > $O/kbench_conv3x3.txt 2>&1; cat $O/kbench_conv3x3.txt
( timeout 120 tools/kbench 256 1 224 20 "dp_project_update" ) > $O/kbench_update_b256.txt 2>&1; cat $O/kbench_update_b256.txt
( timeout 900 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -40 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -15 $O/pytest_gpu.log
( timeout 1200 python bench.py --whole-attack ) > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err; echo "whole-attack rc=$?" | tee -a $O/rc.txt
cat $O/bench_whole_attack.json | head -c 4000; echo
