#!/bin/bash
# Round 5, GPU call 2: the deeper-prefetch / rotated-barrier variants of the 1x1 MFMA kernel on every shape (kbench), SQ + cache
# counters on three shapes, parity of the folded graph (GroupNorm in the conv staging, adds in the conv epilogue) on
# ResNetV2-50, and the headline step A/B: round-4 routes / all 1x1 on the new kernel / + folded graph; trace of the last.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05b; mkdir -p $O
( DP_C1_VARIANTS=0,16,8,24 DP_C1_PLAIN_ONLY=1 timeout 300 tools/kbench 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_variants.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
( DP_C1_VARIANTS=0 timeout 300 tools/kbench 512 1 224 20 conv1x1 ) > $O/kbench_conv1x1_fold_res.txt 2>&1
python - $O/kbench_conv1x1_variants.txt <<'PY'
import re, sys, collections
t = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x.* variant\s+(\d+) plain\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m: t.setdefault((m.group(1), m.group(2), m.group(3)), {})[int(m.group(4))] = (float(m.group(5)), float(m.group(6)))
for k, v in t.items(): print("%5s->%5s @%2s " % k + "  ".join("v%-2d %.3f ms %5.1f TF" % (a, b[0], b[1]) for a, b in v.items()))
PY
grep -E "fold|res" $O/kbench_conv1x1_fold_res.txt | cut -c1-150
SH="256:1024:14,1024:256:14,64:256:56,256:64:56"
( cd /tmp; DP_C1_SHAPES=$SH DP_C1_VARIANTS=0,16 DP_C1_PLAIN_ONLY=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc1 -o kb -- $R/tools/kbench 512 1 224 2 conv1x1 > /dev/null 2> $R/$O/pmc1.err )
( cd /tmp; DP_C1_SHAPES=$SH DP_C1_VARIANTS=0,16 DP_C1_PLAIN_ONLY=1 timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$O/pmc2 -o kb -- $R/tools/kbench 512 1 224 2 conv1x1 > /dev/null 2> $R/$O/pmc2.err )
( cd /tmp; DP_C1_SHAPES=$SH DP_C1_VARIANTS=0 DP_C1_PLAIN_ONLY=1 timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $R/$O/pmc3 -o kb -- $R/tools/kbench 512 1 224 2 conv1x1 > /dev/null 2> $R/$O/pmc3.err )
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_conv1x1" in row["Kernel_Name"]:
            key = (row["Kernel_Name"].split("(")[0][-34:], row["Grid_Size"])
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/sq_counters_conv1x1.txt", "w") as out:
    for key, ctrs in sorted(agg.items()):
        wc = sum(ctrs.get("SQ_WAVE_CYCLES", [0])) / max(1, len(ctrs.get("SQ_WAVE_CYCLES", [])))
        line = "== %s grid %s" % key
        print(line); out.write(line + "\n")
        for c, v in sorted(ctrs.items()):
            m = sum(v) / len(v)
            line = "   %-28s %.4g%s" % (c, m, ("  (%.1f%% of wave cycles)" % (100 * m / wc)) if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else "")
            print(line); out.write(line + "\n")
PY
tail -2 $O/pmc1.err $O/pmc2.err $O/pmc3.err
find $O/pmc1 $O/pmc2 $O/pmc3 -name "*.csv" -size +2M -delete 2>/dev/null
( timeout 900 python -m pytest tests/test_fold_gpu.py tests/test_kernels_gpu.py -m gpu -q -rs -p no:cacheprovider -k "fold or conv1x1 or conv3x3" 2>&1 | tail -15 ) > $O/pytest_fold.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -6 $O/pytest_fold.log
run_bench() {  # name, args...
  name=$1; shift
  ( timeout 400 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("gn_fold"), {k: v for k, v in d["config"].get("conv1x1", {}).items() if k in ("mode", "fwd", "bwd")})
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
run_bench r4routes --conv1x1 table --gn-fold off
run_bench mfma_nofold --conv1x1 mfma --gn-fold off
run_bench mfma_fold --conv1x1 mfma --gn-fold on
run_bench table_fold --conv1x1 table --gn-fold on
run_bench r4routes_2 --conv1x1 table --gn-fold off
run_bench mfma_fold_2 --conv1x1 mfma --gn-fold on
( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fold -o bench -- python $R/bench.py --conv1x1 mfma --gn-fold on --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline > $R/$O/prof_bench_fold.json 2> $R/$O/prof_fold.err ); echo "prof fold rc=$?" | tee -a $O/rc.txt
python scripts/rocpd_stats.py $(ls $O/prof_fold/*kernel_trace.csv | head -1) --timed-steps 3 --top 70 > $O/kernel_stats_timed_fold.txt 2> $O/kernel_stats_fold.err
find $O/prof_fold -name "*.csv" -size +1M -delete
head -40 $O/kernel_stats_timed_fold.txt | cut -c1-170
