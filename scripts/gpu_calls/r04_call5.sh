#!/bin/bash
# Round 4, GPU call 5: the MFMA convolution with the explicit k-step pipeline (go / no-go re-measured, SQ counters);
# the GPU backbone against transformers' BiT in fp64.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04e; mkdir -p $O
( timeout 200 python scripts/conv3x3_vs_miopen.py 512 ) > $O/conv3x3_vs_miopen_n512.json 2> $O/conv3x3.err; cat $O/conv3x3_vs_miopen_n512.json
( timeout 100 python scripts/conv3x3_vs_miopen.py 128 ) > $O/conv3x3_vs_miopen_n128.json 2>> $O/conv3x3.err; cat $O/conv3x3_vs_miopen_n128.json
( timeout 100 tools/kbench 512 1 224 10 "conv3x3" ) > $O/kbench_conv3x3.txt 2>&1; cat $O/kbench_conv3x3.txt
( cd /tmp; timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc -o kb -- $R/tools/kbench 512 1 224 2 "conv3x3" > /dev/null 2> $R/$O/pmc.err )
( cd /tmp; timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc2 -o kb -- $R/tools/kbench 512 1 224 2 "conv3x3" > /dev/null 2> $R/$O/pmc2.err )
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
find $O/pmc $O/pmc2 -name "*.csv" -size +2M -delete 2>/dev/null
( timeout 600 python -m pytest tests/test_backbone_vs_hf_bit_gpu.py tests/test_kernels_gpu.py -m gpu -q -s -rs -p no:cacheprovider -k "transformers_bit or conv3x3 or apply_fwd_exact" 2>&1 | grep -v "mask size" | tail -12 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -8 $O/pytest_subset.log
