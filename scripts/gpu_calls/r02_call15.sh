#!/bin/bash
# Round 2, GPU call 15: route / tuned-solution probes for the GEMM batches of the other single-GPU configs
# (configs[3]: 64 @224, configs[0]: 32 @224, configs[2]: 64 @384).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
for spec in "64 224" "32 224" "64 384"; do
  set -- $spec
  ( DORPATCH_TUNABLEOP=0 timeout 400 python scripts/tunableop_probe.py --n $1 --size $2 --csv $O/tunableop_raw_n$1_$2.csv --max-ms 600 --iters 30 ) > $O/tunableop_probe_n$1_$2.jsonl 2> $O/tunableop_n$1_$2.err; echo "probe n=$1 size=$2 rc=$?" | tee -a $O/rc.txt
done
cat $O/rc.txt; for f in $O/*.jsonl; do tail -1 $f | cut -c1-200; done; ls -la $O
