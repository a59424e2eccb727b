#!/bin/bash
# Round 6, GPU call 18: counters of the Winograd kernel (tools/kbench conv3x3, variant 1000, N = 512): where the waves' cycles
# go (SQ), how busy the matrix pipe is, LDS bank conflicts, and its HBM traffic (WRITE_SIZE and FETCH_SIZE in separate passes).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06t; mkdir -p $O
cd /tmp
pass() {  # name, counters...
  name=$1; shift
  DP_C3_VARIANTS=1000,1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -o kb -- $R/tools/kbench 512 1 224 2 conv3x3 > $O/kbench_under_pmc_$name.txt 2> $O/pmc_$name.err; echo "pmc $name rc=$?" | tee -a $O/rc.txt
}
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
pass wr WRITE_SIZE
pass rd FETCH_SIZE
python - $O <<'PY'
import csv, glob, sys, collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(O+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        n=row["Kernel_Name"]
        if "k_conv3x3_wino" in n or "k_conv3x3_mfma" in n:
            key=n.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O+"/counters_conv3x3_wino_vs_direct.txt","w") as out:
    for k in sorted(agg):
        v=agg[k]
        line=k+"\n   "+"  ".join("%s=%.4g" % (c, sum(x)/len(x)) for c,x in sorted(v.items()))
        wc=sum(v["SQ_WAVE_CYCLES"])/len(v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else 0
        if wc: line+="\n   of wave cycles: "+"  ".join("%s=%.1f%%" % (c, 100*sum(x)/len(x)/wc) for c,x in sorted(v.items()) if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE"))
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE"):
            line+="\n   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) = %.1f%%" % (100*(sum(v["SQ_VALU_MFMA_BUSY_CYCLES"])/len(v["SQ_VALU_MFMA_BUSY_CYCLES"]))/((sum(v["GRBM_GUI_ACTIVE"])/len(v["GRBM_GUI_ACTIVE"]))/8*1024))
        if v.get("WRITE_SIZE") and v.get("FETCH_SIZE"):
            line+="\n   HBM bytes per launch = WRITE_SIZE KiB x 1024 + 2 x FETCH_SIZE KiB x 1024 = %.4g" % (1024*(sum(v["WRITE_SIZE"])/len(v["WRITE_SIZE"]))+2048*(sum(v["FETCH_SIZE"])/len(v["FETCH_SIZE"])))
        print(line); out.write(line+"\n")
PY
find $O -name "*.csv" -size +2M -delete
