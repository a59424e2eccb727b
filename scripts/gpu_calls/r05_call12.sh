#!/bin/bash
# Round 5, GPU call 12: dp_stem_conv_fwd (stem 7x7/2 on the matrix cores) — kbench, parity, against MIOpen, the headline
# step with it on / off on one box; then a one-stream kernel trace of the tree (what is left outside own kernels).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05l; mkdir -p $O
( timeout 200 tools/kbench 512 1 224 20 stemconv ) > $O/kbench_stemconv.txt 2>&1; echo "kbench rc=$?" | tee -a $O/rc.txt
cut -c1-200 $O/kbench_stemconv.txt
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rs -x -p no:cacheprovider -k "stem" 2>&1 | tail -15 ) > $O/pytest_stem.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -4 $O/pytest_stem.log
python - > $O/stem_vs_miopen.txt 2>&1 <<'PY'
import torch, torch.nn.functional as F
from dorpatch_amd import ops
def timed(fn, iters=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters
for N in (512, 128, 64, 32):
    x = torch.randn(N, 3, 224, 224, device="cuda"); w = torch.randn(64, 3, 7, 7, device="cuda") / 12
    wt = ops.pack_stem_weights(w)
    want = F.conv2d(x, w, stride=2, padding=3); got = ops.stem_conv_fwd(x, wt)
    print("N=%d miopen %.4f ms  own %.4f ms  max rel diff %.2e" % (N, timed(lambda: F.conv2d(x, w, stride=2, padding=3)), timed(lambda: ops.stem_conv_fwd(x, wt)), float((got - want).abs().max() / want.abs().max())), flush=True)
PY
cat $O/stem_vs_miopen.txt
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  ( env $envs timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-sweep --no-cpu-baseline --no-pmc --no-update-roofline --no-conv-roofline ) > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?" | tee -a $O/rc.txt
  python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]; print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("step_tflops"), d["config"].get("streams"), d["config"].get("conv3x3"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  tail -1 $O/bench_$name.err | cut -c1-300
}
run stem_on DORPATCH_STEM_CONV=on
run stem_off DORPATCH_STEM_CONV=off
run stem_on_b DORPATCH_STEM_CONV=on
run stem_off_b DORPATCH_STEM_CONV=off
run stem_on_s3 DORPATCH_STEM_CONV=on --streams 3
( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_s1 -o bench -- python $R/bench.py --streams 1 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_s1.json 2> $R/$O/prof_s1.err ); echo "prof s1 rc=$?" | tee -a $O/rc.txt
python scripts/rocpd_stats.py $(ls $O/prof_s1/*kernel_trace.csv | head -1) --timed-steps 3 --top 80 > $O/kernel_stats_timed_headline_streams1.txt 2> $O/kernel_stats_s1.err
find $O/prof_s1 -name "*.csv" -size +1M -delete
head -45 $O/kernel_stats_timed_headline_streams1.txt | cut -c1-170
