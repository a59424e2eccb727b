#!/bin/bash
# Round 6, GPU call 27: counters of k_conv1x1_mfma on its slowest (64 -> 256 @56x56) and one of its fastest (1024 -> 256 @14x14)
# shapes, plain and fold + add, N = 512: where the waves' cycles go, how busy the matrix pipe is, cache / fabric requests.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06z_pmc; mkdir -p $O
cd /tmp
pass() {  # name, counters...
  name=$1; shift
  DP_C1_MODES=0,3 DP_C1_SHAPES=64:256:56,1024:256:14 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -o kb -- $R/tools/kbench 512 1 224 2 conv1x1 > $O/kbench_under_pmc_$name.txt 2> $O/pmc_$name.err; echo "pmc $name rc=$?" | tee -a $O/rc.txt
}
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
pass vmem SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
pass ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
pass wr WRITE_SIZE
pass rd FETCH_SIZE
python - $O <<'PY'
import csv, glob, sys, collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(O+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        n=row["Kernel_Name"]
        if "k_conv1x1_mfma" in n:
            key=n.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")+"  grid %s" % row.get("Grid_Size","?")
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O+"/counters_conv1x1.txt","w") as out:
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
cat $O/rc.txt
