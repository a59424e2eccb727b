#!/bin/bash
# Round 2, GPU call 4: library-route table incl. the subsampled downsample shapes (hipBLASLt default vs rocBLAS),
# store-flavour / copy calibration, kernel micro-benchmarks, deterministic-mode A/B, 2-rank product check on real
# kernels, GPU suite.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
( timeout 150 python scripts/conv1x1_table.py ) > $O/conv1x1_table_default.jsonl 2> $O/c1.err; echo "table rc=$?" | tee -a $O/rc.txt
( timeout 150 python scripts/conv1x1_table.py --blas cublas ) > $O/conv1x1_table_rocblas.jsonl 2> $O/c2.err; echo "table rocblas rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 calib ) > $O/kbench_calib.txt 2>&1; echo "kbench calib rc=$?" | tee -a $O/rc.txt
( timeout 100 tools/kbench 64 32 224 20 maxpool ) > $O/kbench_maxpool.txt 2>&1; echo "kbench maxpool rc=$?" | tee -a $O/rc.txt
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 scripts/two_rank_check.py ) > $O/two_rank_check.json 2> $O/two_rank_check.err; echo "2rank rc=$?" | tee -a $O/rc.txt
( time timeout 900 python -m pytest tests -m gpu -q -rf -s --durations=5 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_det.json 2> $O/bench_det.err; echo "bench det rc=$?" | tee -a $O/rc.txt
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 --deterministic off ) > $O/bench_nondet.json 2> $O/bench_nondet.err; echo "bench nondet rc=$?" | tee -a $O/rc.txt
for c in 2 3; do ( timeout 200 python bench.py --config $c --no-cpu-baseline --no-pmc --no-sweep --steps 10 --warmup 3 ) > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?" | tee -a $O/rc.txt; done
cat $O/rc.txt; cat $O/kbench_calib.txt; cat $O/kbench_maxpool.txt | tail -3; cut -c1-700 $O/two_rank_check.json; tail -3 $O/two_rank_check.err; tail -6 $O/pytest_gpu.log
for f in bench_det bench_nondet bench_cfg2 bench_cfg3; do cut -c1-200 $O/$f.json; done
python - <<'PY'
import json,sys
for name in ("default","rocblas"):
    try:
        rows=[json.loads(l) for l in open("gpurun_out/r02d/conv1x1_table_%s.jsonl"%name) if l.startswith("{")]
    except Exception as e:
        print(name, e); continue
    tot_g=sum(r["gemm_ms"]*r["per_forward"] for r in rows); tot_m=sum(r["miopen_ms"]*r["per_forward"] for r in rows)
    best=sum(min(r["gemm_ms"],r["miopen_ms"])*r["per_forward"] for r in rows)
    print(name, len(rows), "rows; gemm-only %.2f ms, miopen-only %.2f ms, best-of %.2f ms"%(tot_g,tot_m,best))
    for r in rows:
        if (r["C"],r["O"]) in ((256,512),(512,1024),(1024,2048)): print("  ",r)
PY
