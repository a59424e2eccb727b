#!/bin/bash
# Round 6, GPU call 7: what the failure sweep's forwards consist of (kernel trace of a bench run whose timed region is 1 step +
# the sweep), one and two streams; sweep time with micro-batches of 512 and 1024 rows.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06g; mkdir -p $O
for s in 1 2; do
( cd /tmp; timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sweep_s$s -o bench -- python $R/bench.py --streams $s --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_sweep_s$s.json 2> $R/$O/prof_sweep_s$s.err ); echo "prof sweep s$s rc=$?" | tee -a $O/rc.txt
python - $O/prof_sweep_s$s $O/kernel_stats_sweep_s$s.txt <<'PY'
import csv,glob,sys,collections
d,out=sys.argv[1],sys.argv[2]
path=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[]
for r in csv.DictReader(open(path)):
    rows.append((r["Kernel_Name"],int(r["Start_Timestamp"]),int(r["End_Timestamp"])))
rows.sort(key=lambda r:r[1])
# the sweep = everything after the LAST k_project_update launch's predecessor step... take the window after the last k_cw_loss
last=max(i for i,r in enumerate(rows) if "k_cw_loss" in r[0])
# skip the rest of that step: start at the first k_argmax after it minus its forward: use first k_apply_fwd after `last`
start=next(i for i in range(last,len(rows)) if "k_apply_fwd" in rows[i][0])
sel=rows[start:]
agg=collections.defaultdict(lambda:[0,0.0])
for n,a,b in sel:
    agg[n][0]+=1; agg[n][1]+=(b-a)/1e6
wall=(max(r[2] for r in sel)-sel[0][1])/1e6
tot=sum(v[1] for v in agg.values())
with open(out,"w") as f:
    f.write("# sweep window: %.1f ms wall, %.1f ms kernel time, %d dispatches\n"%(wall,tot,len(sel)))
    for n,(c,ms) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:60]:
        f.write("%-110s %6d %10.3f ms %5.2f%% avg %8.2f us\n"%(n[:110],c,ms,100*ms/tot,1e3*ms/c))
print(open(out).read()[:3500])
PY
find $O/prof_sweep_s$s -name "*.csv" -size +1M -delete
done
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
run sweep_mb512 X=1 --steps 3 --warmup 1
run sweep_mb1024 X=1 --steps 3 --warmup 1 --micro-batch 1024
run sweep_mb512_s1 X=1 --steps 3 --warmup 1 --streams 1
