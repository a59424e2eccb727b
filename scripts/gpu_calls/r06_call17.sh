#!/bin/bash
# Round 6, GPU call 17: one-stream kernel traces of the small configurations on the final tree (configs[3]: 1 x 64; the
# reference's 1 x 128; configs[0]: 8 x 4; configs[2]: 384^2).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06s; mkdir -p $O
prof() {  # name, timed steps, args
  name=$1; shift; st=$1; shift
  ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -o bench -- python $R/bench.py "$@" --streams 1 --steps $st --warmup 3 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $R/$O/prof_bench_$name.json 2> $R/$O/prof_$name.err ); echo "prof $name rc=$?" | tee -a $O/rc.txt
  python scripts/rocpd_stats.py $(ls $O/prof_$name/*kernel_trace.csv | head -1) --timed-steps $st --top 90 > $O/kernel_stats_timed_$name.txt 2> $O/kernel_stats_$name.err
  find $O/prof_$name -name "*.csv" -size +1M -delete
}
prof cfg3 10 --config 3
prof b1s128 10 --batch 1 --samples 128
prof cfg0 10 --config 0
prof cfg2 5 --config 2
head -50 $O/kernel_stats_timed_cfg3.txt | cut -c1-150
