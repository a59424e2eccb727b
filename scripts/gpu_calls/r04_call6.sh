#!/bin/bash
# Round 4, GPU call 6: the generalized MFMA 3x3 convolution on ResNetV2-50's four shapes vs MIOpen (N = 512, 128, 32),
# its parity tests, the backbone parity suites with the route in place, and the headline step with the route off /
# table (64@56 only) / on (all four shapes).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04f; mkdir -p $O
( timeout 300 python scripts/conv3x3_vs_miopen.py 512 128 32 ) > $O/conv3x3_vs_miopen.jsonl 2> $O/conv3x3.err; cat $O/conv3x3_vs_miopen.jsonl | cut -c1-420
( timeout 100 tools/kbench 512 1 224 10 "conv3x3" ) > $O/kbench_conv3x3.txt 2>&1; cat $O/kbench_conv3x3.txt
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_backbone_parity_gpu.py tests/test_headline_parity_gpu.py tests/test_backbone_vs_hf_bit_gpu.py -m gpu -q -rs -p no:cacheprovider -k "conv3x3 or backbone or headline or transformers or fresh or reproduc" 2>&1 | grep -v "mask size" | tail -12 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/rc.txt
tail -8 $O/pytest_subset.log
for mode in off table on; do
  ( timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-sweep --conv3x3 $mode ) > $O/bench_conv3x3_$mode.json 2> $O/bench_conv3x3_$mode.err; echo "bench $mode rc=$?" | tee -a $O/rc.txt
  python - $O/bench_conv3x3_$mode.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], d["value"], "EOT-samples/s", d["ms_per_step"], "ms/step", d["config"]["conv3x3"], d["config"]["backward"]["step_ms"])
PY
done
