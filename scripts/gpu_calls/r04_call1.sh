#!/bin/bash
# Round 4, GPU call 1: full GPU suite on the new tree (retire, wide-lane project_update, XCD-aware apply order, the
# INTEGRATION.md stub), kbench A/B of the touched kernels, the plain multi-rank command on one GPU (gloo, same device),
# a short bench, the whole-attack measurement.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04a; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x -rs -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -25 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -8 $O/pytest_gpu.log
( timeout 120 tools/kbench 64 32 224 20 "dp_apply_fwd (" ) > $O/kbench_apply_order.txt 2>&1
( timeout 120 tools/kbench 64 1 224 20 "dp_project_update" ) > $O/kbench_update_b64.txt 2>&1
( timeout 120 tools/kbench 256 1 224 20 "dp_project_update" ) > $O/kbench_update_b256.txt 2>&1
cat $O/kbench_apply_order.txt $O/kbench_update_b64.txt $O/kbench_update_b256.txt
( timeout 400 python bench.py --gpus 2 --same-device --backend gloo --batch 8 --samples 8 --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --no-pmc ) > $O/bench_plain_2rank_gloo.json 2> $O/bench_plain_2rank_gloo.err; echo "plain 2-rank rc=$?" | tee -a $O/rc.txt
( timeout 60 python bench.py --gpus 4 --backend nccl --steps 1 --warmup 0 ) > $O/bench_plain_4gpu_refused.out 2>&1; echo "plain 4-gpu on a 1-gpu box rc=$? (2 expected)" | tee -a $O/rc.txt
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?" | tee -a $O/rc.txt
cat $O/bench_short.json | head -c 3000; echo
( timeout 1200 python bench.py --whole-attack ) > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err; echo "whole-attack rc=$?" | tee -a $O/rc.txt
cat $O/bench_whole_attack.json | head -c 4000; echo
tail -5 $O/bench_whole_attack.err
