#!/bin/bash
# Round 4, GPU call 9: HBM traffic of dp_project_update (PMC passes on the kbench replay, 256 images), the N = 1 point of
# the configs[3] strong-scaling mode, the whole attack on the final tree.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04i; mkdir -p $O
for ctr in WRITE_SIZE FETCH_SIZE; do
  ( cd /tmp; timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$O/pmc_$ctr -o kb -- $R/tools/kbench 256 1 224 2 "dp_project_update stage 0" > /dev/null 2> $R/$O/pmc_$ctr.err )
done
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_project_update" in row["Kernel_Name"] or "k_calib_update" in row["Kernel_Name"]:
            agg[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/pmc_project_update.txt", "w") as out:
    for k, v in agg.items():
        w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024 if v.get("WRITE_SIZE") else float("nan")
        f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024 if v.get("FETCH_SIZE") else float("nan")
        line = "%-42s WRITE_SIZE %.1f MB  FETCH_SIZE %.1f MB (x2 on gfx950 = %.1f MB)  traffic %.1f MB  (algorithmic, stage 0, 256 images: 924.8 MB = 719.3 read + 205.5 written)" % (k, w / 1e6, f / 1e6, 2 * f / 1e6, (w + 2 * f) / 1e6)
        print(line); out.write(line + "\n")
PY
find $O -name "*.csv" -size +1M -delete 2>/dev/null
( timeout 300 python bench.py --config 3 --scaling strong --gpus 1 --steps 10 --warmup 3 --no-sweep --no-pmc --no-cpu-baseline ) > $O/bench_cfg3_strong_n1.json 2> $O/bench_cfg3_strong_n1.err; echo "cfg3 strong rc=$?" | tee -a $O/rc.txt
python - $O/bench_cfg3_strong_n1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["scaling"], d["config"]["workload"][:120])
PY
( timeout 1300 python bench.py --whole-attack ) > $O/bench_whole_attack.json 2> $O/bench_whole_attack.err; echo "whole-attack rc=$?" | tee -a $O/rc.txt
cat $O/bench_whole_attack.json | head -c 3500; echo; grep "whole attack" $O/bench_whole_attack.err
