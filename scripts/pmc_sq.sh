#!/bin/bash
# SQ counters of the apply kernels (one rocprofv3 --pmc pass, kernel-trace only): where do the waves' cycles go?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${PMC_OUT:-r03j}; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc -o kb -- $R/tools/kbench 64 32 224 2 "affine" > $O/kbench_under_pmc.txt 2> $O/pmc.err
python - $O <<'PY'
import csv, glob, sys, collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(O+"/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name=row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::","")
        if "k_apply" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out=open(O+"/sq_counters_apply_kernels.txt","w")
for k,v in agg.items():
    line=k+"\n   "+"  ".join("%s=%.3g" % (c, sum(x)/len(x)) for c,x in sorted(v.items()))
    wc=sum(v["SQ_WAVE_CYCLES"])/len(v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else 0
    if wc: line+="\n   fractions of wave cycles: " + "  ".join("%s=%.1f%%" % (c, 100*sum(x)/len(x)/wc) for c,x in sorted(v.items()) if c!="SQ_WAVE_CYCLES" and c!="SQ_INSTS_VALU")
    print(line); out.write(line+"\n")
PY
find $O/pmc -name "*.csv" -size +4M -delete
