#!/bin/bash
# NOTE: records an experiment whose code was NOT kept (the kbench variants / switches it names are described in DESIGN.md section 9 and in the profiles it wrote); it does not run on the committed tree as is.
# Round 6, GPU call 26 (experiment): do the two workgroups of a CU / the CUs of the chip run k_conv1x1_mfma in lock-step (all in
# their K loops, then all in their epilogues)?  First-round workgroups start late by a slot-dependent delay; N = 512.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06z; mkdir -p $O
for st in 0 17 33 65 18 34 66 19 35 67 0; do
  DP_C1_STAGGER=$st DP_C1_MODES=0,3 DP_C1_SHAPES=64:256:56,128:512:28,256:1024:14,256:64:56,1024:256:14 timeout 200 tools/kbench 512 1 224 10 conv1x1 2>&1 | sed "s/^/stagger $st /" >> $O/kbench_stagger.txt
done
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/r06z/kbench_stagger.txt'):
    m=re.match(r'stagger (\d+) dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x.*variant\s+\d+ (\S+)\s+([\d.]+) ms',l)
    if m: d.setdefault((m.group(2),m.group(3),m.group(4),m.group(5)),collections.OrderedDict()).setdefault(m.group(1),[]).append(m.group(6))
for k,v in d.items(): print(k, ' '.join('%s:%s'%(s,'/'.join(t)) for s,t in v.items()))
PY
