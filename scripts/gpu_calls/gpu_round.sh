#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel trace of the bench.
# Usage (build box): gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showproductname > $O/box.txt 2>&1; lscpu | head -20 >> $O/box.txt; nproc >> $O/box.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
timeout 1500 python bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-1} > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?" | tee -a $O/rc.txt
ls -R $O/prof | head -30
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/bench.json; tail -5 $O/bench.err
