#!/bin/bash
# NOTE: records an experiment whose code was NOT kept (the kbench variants / switches it names are described in DESIGN.md section 9 and in the profiles it wrote); it does not run on the committed tree as is.
# Round 6, GPU call 20: k_conv1x1_mfma A/B on one box — residual fragments requested 2 / 3 ahead of the store (DP_C1_RES_DEPTH
# 3 / 4 against 2) and the staging's loads past the end of K collapsed onto one cache line (DP_C1_TAIL_COLLAPSE), N = 512.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06w; mkdir -p $O
for rep in 1 2; do
for v in base rd3 rd4 tc tc_rd4; do
  DP_C1_MODES=0,1,2,3 DP_C1_SHAPES=64:256:56,128:512:28,256:1024:14,512:2048:7,64:64:56,256:64:56,512:128:28,1024:256:14 timeout 300 tools/kbench_$v 512 1 224 10 conv1x1 > $O/kbench_${v}_$rep.txt 2>&1; echo "$v rc=$?" >> $O/rc.txt
done
done
python - <<'PY'
import re,glob,collections
d=collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/r06w/kbench_*_?.txt')):
    v=re.search(r'kbench_(.*)_(\d)\.txt',f).group(1)
    for l in open(f):
        m=re.match(r'dp_conv1x1_fwd\s+(\d+)->\s*(\d+) @\s*(\d+)x.*variant\s+\d+ (\S+)\s+([\d.]+) ms',l)
        if m:
            k=(int(m.group(1)),int(m.group(2)),int(m.group(3)),m.group(4))
            d[k].setdefault(v,[]).append(float(m.group(5)))
vs=['base','rd3','rd4','tc','tc_rd4']
print('%-28s'%'shape'+''.join('%16s'%v for v in vs))
for k in d:
    print('%-28s'%str(k)+''.join('%16s'%('/'.join('%.4f'%x for x in d[k].get(v,[]))) for v in vs))
PY
