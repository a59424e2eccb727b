#!/bin/bash
# Round 6, GPU call 15: the fold coefficients of dp_conv1x1_fwd read per item from global memory (round 5, default) vs from a per-chunk LDS table
# (DP_DEBUG_CONV1X1_VARIANT bit 9): every folded 1x1 shape at N = 512 / 64, the step, the sweep, configs[3].
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06q; mkdir -p $O
timeout 300 python /root/repo/../nonexistent 2>/dev/null; echo skip-pytest
for n in 512 64; do
  DP_C1_VARIANTS=0,512 DP_C1_MODES=1,3 timeout 600 tools/kbench $n 1 224 20 conv1x1 > $O/kbench_conv1x1_fold_modes_n$n.txt 2>&1; echo "kbench n=$n rc=$?" | tee -a $O/rc.txt
done
python - $O/kbench_conv1x1_fold_modes_n512.txt <<'PY'
import re,sys,collections
rows=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.match(r'dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x\s*\d+ N=(\d+) variant\s+(\d+) (\S+)\s+([\d.]+) ms\s+([\d.]+) TFLOP',l)
    if m:
        C,O,S,N,v,mode,ms,tf=m.groups(); rows.setdefault((C,O,S,mode),{})[int(v)]=float(tf)
print("TFLOP/s: fold coefficients per item from global memory (0) vs from an LDS table (512)")
for k,v in rows.items(): print("%5s->%5s @%2s %-6s"%k, " ".join("%6.1f"%v.get(x,0) for x in (0,512)))
PY
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 900 python bench.py "$@" --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" >> $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("collect_failure_sweep_ms"), d.get("value_with_sweep_amortised"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in 1 2; do
run fold_global_$rep X=1 --steps 10 --warmup 2 --no-sweep
run fold_table_$rep DORPATCH_BENCH_DEBUG_SET=5=512 --steps 10 --warmup 2 --no-sweep
done
run sweep_fold_global X=1 --steps 2 --warmup 1
run sweep_fold_table DORPATCH_BENCH_DEBUG_SET=5=512 --steps 2 --warmup 1
run cfg3_fold_global X=1 --config 3 --steps 20 --warmup 3 --no-sweep
run cfg3_fold_table DORPATCH_BENCH_DEBUG_SET=5=512 --config 3 --steps 20 --warmup 3 --no-sweep
