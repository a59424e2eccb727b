#!/bin/bash
# Round 6, GPU call 16: GroupNorm kernels that FIT BESIDE the MFMA kernels — 256-thread streaming workgroups (36 - 40 VGPRs: one
# wave per SIMD next to two MFMA waves; the register-resident kernels need a whole CU's register file and only run where an
# MFMA workgroup has left) — in the step, by stream count.  DP_DEBUG_GN_VARIANT: 1 backward, 2 forward / statistics, 3 both.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06r; mkdir -p $O
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
run base_s2 X=1 --steps 8 --warmup 2 --no-sweep
run gnbwd_s2 DORPATCH_BENCH_DEBUG_SET=7=1 --steps 8 --warmup 2 --no-sweep
run gnboth_s2 DORPATCH_BENCH_DEBUG_SET=7=3 --steps 8 --warmup 2 --no-sweep
run gnbwd_s3 DORPATCH_BENCH_DEBUG_SET=7=1 --steps 8 --warmup 2 --no-sweep --streams 3
run gnboth_s3 DORPATCH_BENCH_DEBUG_SET=7=3 --steps 8 --warmup 2 --no-sweep --streams 3
run gnbwd_s4 DORPATCH_BENCH_DEBUG_SET=7=1 --steps 8 --warmup 2 --no-sweep --streams 4
run gnboth_s4 DORPATCH_BENCH_DEBUG_SET=7=3 --steps 8 --warmup 2 --no-sweep --streams 4
run gnbwd_s1 DORPATCH_BENCH_DEBUG_SET=7=1 --steps 5 --warmup 2 --no-sweep --streams 1
run base_s2_b X=1 --steps 8 --warmup 2 --no-sweep
( cd /tmp; DORPATCH_BENCH_DEBUG_SET=7=3 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_s2 -o bench -- python $R/bench.py --streams 2 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_s2.json 2> $R/$O/prof_s2.err ); echo "prof rc=$?" | tee -a $O/rc.txt
python scripts/rocpd_stats.py $(ls $O/prof_s2/*kernel_trace.csv | head -1) --timed-steps 3 --top 30 > $O/kernel_stats_timed_gnstream_streams2.txt 2> $O/kernel_stats.err
find $O/prof_s2 -name "*.csv" -size +1M -delete
head -16 $O/kernel_stats_timed_gnstream_streams2.txt | cut -c1-150
