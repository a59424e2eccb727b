#!/bin/bash
# Round 5, GPU call 5: the whole GPU suite on the folded graph (regressions?), smoke, the default bench line with
# roofline_conv / step_tflops, per-shape 1x1 timing against the libraries at batches 512 / 128 / 64 / 32 (route table),
# the other single-GPU configs with the fold on / off.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05e; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -rs -x -p no:cacheprovider 2>&1 | grep -v "mask size" | tail -40 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -25 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -2 $O/smoke.log
( timeout 900 python bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --no-pmc ) > $O/bench_default_short.json 2> $O/bench_default_short.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - $O/bench_default_short.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(d["value"], d["ms_per_step"], d.get("step_tflops"), d.get("step_frac_of_peak")); print(json.dumps(d.get("roofline_conv"))[:1500])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
tail -3 $O/bench_default_short.err
( timeout 900 python scripts/conv1x1_vs_lib.py 512 128 64 32 ) > $O/conv1x1_vs_lib.jsonl 2> $O/conv1x1_vs_lib.err; echo "vs_lib rc=$?" | tee -a $O/rc.txt
python - $O/conv1x1_vs_lib.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "dir" not in d: print(d); continue
    print(d["N"], d["dir"], d["C"], d["O"], d["HW"], " ".join("%s=%.3f" % kv for kv in d["ms"].items()), "lib/mfma", d["lib_over_mfma"])
PY
for cfg in 2 3 0; do for fold in on off; do
  ( timeout 300 python bench.py --config $cfg --gn-fold $fold --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline --no-conv-roofline --no-update-roofline ) > $O/bench_cfg${cfg}_fold_$fold.json 2> $O/bench_cfg${cfg}_fold_$fold.err; echo "cfg$cfg $fold rc=$?" | tee -a $O/rc.txt
  python - $O/bench_cfg${cfg}_fold_$fold.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done; done
( timeout 300 python bench.py --batch 1 --samples 128 --gn-fold on --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline --no-conv-roofline --no-update-roofline ) > $O/bench_b1s128_fold_on.json 2> /dev/null
( timeout 300 python bench.py --batch 1 --samples 128 --gn-fold off --steps 20 --warmup 5 --no-sweep --no-pmc --no-cpu-baseline --no-conv-roofline --no-update-roofline ) > $O/bench_b1s128_fold_off.json 2> /dev/null
python - $O/bench_b1s128_fold_on.json $O/bench_b1s128_fold_off.json <<'PY'
import json,sys
for p in sys.argv[1:]:
    try:
        d=[json.loads(l) for l in open(p).read().strip().splitlines() if l.startswith("{")][-1]; print(p.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(p, "unreadable", e)
PY
