#!/bin/bash
# Round 6, GPU call 19: the small configurations as two half-size micro-batches on two streams against one micro-batch
# (configs[3]: 64 rows; the reference's default 1 x 128; configs[0]: 32 rows), interleaved A/B on one box; plus a smoke run
# of the split source tree (kernel-family files) and the kernel tests.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06v; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1)" | tee -a $O/rc.txt; }
for rep in 1 2; do
  run cfg3_one_$rep --config 3
  run cfg3_mb32_s2_$rep --config 3 --micro-batch 32 --streams 2
  run b1s128_one_$rep --batch 1 --samples 128
  run b1s128_mb64_s2_$rep --batch 1 --samples 128 --micro-batch 64 --streams 2
  run cfg0_one_$rep --config 0
  run cfg0_mb16_s2_$rep --config 0 --micro-batch 16 --streams 2
done
run cfg1_mb256_s2 --config 1 --micro-batch 256 --streams 2
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_kernels.log
cat $O/rc.txt
