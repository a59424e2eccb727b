#!/bin/bash
# Round 4, GPU call 4: placement backward with the branch-free gather (A/B against the round-3 loop), the ceiling of
# dp_project_update's access pattern, the placement GPU tests, the whole attack with warmed shapes.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04d; mkdir -p $O
( timeout 200 tools/kbench 64 32 224 10 "affine" ) > $O/kbench_affine.txt 2>&1; cat $O/kbench_affine.txt
( timeout 120 tools/kbench 256 1 224 20 "dp_project_update" ) > $O/kbench_update_b256.txt 2>&1; cat $O/kbench_update_b256.txt
( timeout 600 python -m pytest tests/test_placement_gpu.py tests/test_kernels_gpu.py -m gpu -q -rs -p no:cacheprovider -k "placement or affine or gather or project_update or full_size or walk or adjoint or slab" 2>&1 | grep -v "mask size" | tail -12 ) > $O/pytest_placement.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -6 $O/pytest_placement.log
( cd /tmp; timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/pmc -o kb -- $R/tools/kbench 64 32 224 2 "dp_apply_affine_bwd (+" > /dev/null 2> $R/$O/pmc.err )
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for path in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_apply_affine_bwd" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/sq_counters_affine_bwd.txt", "w") as out:
    wc = sum(agg.get("SQ_WAVE_CYCLES", [0])) / max(1, len(agg.get("SQ_WAVE_CYCLES", [])))
    for c, v in sorted(agg.items()):
        m = sum(v) / len(v)
        line = "%-24s %.4g%s" % (c, m, ("  (%.1f%% of wave cycles)" % (100 * m / wc)) if wc and c not in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU") else "")
        print(line); out.write(line + "\n")
PY
find $O/pmc -name "*.csv" -size +2M -delete 2>/dev/null
( timeout 1300 python bench.py --whole-attack ) > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err; echo "whole-attack rc=$?" | tee -a $O/rc.txt
cat $O/bench_whole_attack.json | head -c 4000; echo; grep "whole attack" $O/bench_whole_attack.err
