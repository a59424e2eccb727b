#!/bin/bash
# Round 4, GPU call 3: the 224x224 end-metric test; project_update (1-D) and conv (mid-chunk stash) re-measured, SQ
# counters of the conv kernel; whole attack with padded sweep tails.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04c; mkdir -p $O
( timeout 600 python -m pytest tests/test_end_metric_gpu.py tests/test_kernels_gpu.py tests/test_attack_gpu.py -m gpu -q -s -rs -p no:cacheprovider -k "224_through or project_update or conv3x3 or finished or retired or collect_failure" 2>&1 | grep -v "mask size" | tail -30 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -14 $O/pytest_subset.log
( timeout 120 tools/kbench 256 1 224 20 "dp_project_update" ) > $O/kbench_update_b256.txt 2>&1; cat $O/kbench_update_b256.txt
( timeout 120 tools/kbench 64 1 224 20 "dp_project_update" ) > $O/kbench_update_b64.txt 2>&1; cat $O/kbench_update_b64.txt
( timeout 120 tools/kbench 64 1 384 20 "dp_project_update" ) > $O/kbench_update_b64_384.txt 2>&1; cat $O/kbench_update_b64_384.txt
( timeout 200 python scripts/conv3x3_vs_miopen.py 512 ) > $O/conv3x3_vs_miopen_n512.json 2> $O/conv3x3.err; cat $O/conv3x3_vs_miopen_n512.json
( cd /tmp; timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc -o kb -- $R/tools/kbench 512 1 224 2 "conv3x3" > $R/$O/kbench_conv_under_pmc.txt 2> $R/$O/pmc.err )
( cd /tmp; timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc2 -o kb -- $R/tools/kbench 512 1 224 2 "conv3x3" > /dev/null 2> $R/$O/pmc2.err )
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for path in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_conv3x3" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/sq_counters_conv3x3.txt", "w") as out:
    wc = sum(agg.get("SQ_WAVE_CYCLES", [0])) / max(1, len(agg.get("SQ_WAVE_CYCLES", [])))
    for c, v in sorted(agg.items()):
        m = sum(v) / len(v)
        line = "%-28s %.4g%s" % (c, m, ("  (%.1f%% of wave cycles)" % (100 * m / wc)) if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else "")
        print(line); out.write(line + "\n")
PY
tail -3 $O/pmc.err $O/pmc2.err
find $O/pmc $O/pmc2 -name "*.csv" -size +2M -delete 2>/dev/null
( timeout 1200 python bench.py --whole-attack ) > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err; echo "whole-attack rc=$?" | tee -a $O/rc.txt
cat $O/bench_whole_attack.json | head -c 4000; echo
