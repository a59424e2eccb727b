#!/bin/bash
# Round 2, GPU call 19: the sandboxed self-test of the tuned GEMM solutions (cost, verdict), bench, GPU suite.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
( time python - <<'PY'
import time, torch
from dorpatch_amd import conv1x1
t0 = time.time(); ok = conv1x1._selftest_tuned(); print("selftest verdict", ok, "%.1f s" % (time.time() - t0))
print("active", conv1x1.tuned_gemms_active(True), conv1x1.report_tuned())
PY
) > $O/selftest.txt 2>&1; echo "selftest rc=$?" | tee -a $O/rc.txt
( time timeout 400 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
( time timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-pmc --samples 16 ) > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/rc.txt
cat $O/rc.txt; cat $O/selftest.txt | tail -6; cut -c1-200 $O/bench.json; tail -4 $O/bench.err; tail -4 $O/pytest_gpu.log; cut -c1-160 $O/bench_2rank_gloo.json
