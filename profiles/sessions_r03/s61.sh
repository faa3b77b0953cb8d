#!/bin/bash
# round 3, session 22: fir_fft with batched slab reads; per-kernel trace of the cfg4 chain
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s61; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "low_pass or high_pass or equalizer or fir or sinc or cfg4" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python tools/cfgbench.py --only lowpass,eq,applyir,chain > $O/cfg.log 2>&1; grep -v amdgpu $O/cfg.log | tail -7
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4prof -o k -- python $R/bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg4_bench.log 2>&1
f=$(find $O/cfg4prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg4_kernel_stats.csv
rm -rf $O/cfg4prof
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/s61/cfg4_kernel_stats.csv')))
for r in rows[:9]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(4), '%.3f'%(float(r['AverageNs'])/1e6), r['Percentage'])
PY
grep "^{" $O/cfg4_bench.log | tail -1 | cut -c1-400
