#!/bin/bash
# round 3, session 18: per-kernel trace of the cfg4 chain after the copy removal; bench-command profile (60-step stats pass)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s57; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4prof -o k -- python $R/bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg4_bench.log 2>&1
f=$(find $O/cfg4prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg4_kernel_stats.csv
rm -rf $O/cfg4prof
tail -1 $O/cfg4_bench.log | cut -c1-600
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/s57/cfg4_kernel_stats.csv')))
for r in rows[:26]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(4), '%.3f'%(float(r['AverageNs'])/1e6), '%.2f'%(float(r['TotalDurationNs'])/1e6/13), r['Percentage'])
PY
bash tools/profile_round.sh r03 > $O/profile.log 2>&1
cp gpurun_out/profile_r03/summary.json $O/r03_bench_pmc_summary.json 2>/dev/null
cp gpurun_out/profile_r03/kernel_stats.csv $O/r03_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/profile_r03/stats gpurun_out/profile_r03/pmc_*
head -4 $O/r03_bench_kernel_stats.csv | cut -c1-200
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1500
