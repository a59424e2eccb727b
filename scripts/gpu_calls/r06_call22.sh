#!/bin/bash
# Round 6, GPU call 22: where the HOST spends a configs[0] step (32 rows: 245 launches in ~9 ms, 76-88 % GPU-busy) — cProfile of
# bench.py --config 0 over 300 steps.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06y; mkdir -p $O
timeout 600 python -m cProfile -o $O/cfg0.prof bench.py --config 0 --steps 300 --warmup 10 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $O/bench_cfg0_cprofile.json 2> $O/bench_cfg0_cprofile.err
python - <<'PY' > gpurun_out/r06y/cfg0_host_profile.txt 2>&1
import pstats
p=pstats.Stats('gpurun_out/r06y/cfg0.prof')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumulative').print_stats(60)
PY
rm -f $O/cfg0.prof
timeout 300 python bench.py --config 0 --steps 300 --warmup 10 --no-cpu-baseline --no-sweep --no-pmc --no-update-roofline --no-conv-roofline > $O/bench_cfg0_300.json 2> $O/bench_cfg0_300.err
tail -c 600 $O/bench_cfg0_300.err
head -70 $O/cfg0_host_profile.txt | cut -c1-160
