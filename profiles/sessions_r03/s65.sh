#!/bin/bash
# round 3, session 26: inverse transform at the generic sizes (frame buffer + gather path), per-kernel split
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s65; mkdir -p $O
timeout 120 python tools/kbench.py --what istft --iters 10 --batch 256 --sr 96000 --nfft 4096 2>&1 | grep -v amdgpu | tee $O/istft4096.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/tools/kbench.py --what istft --iters 5 --batch 256 --sr 96000 --nfft 4096 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/istft4096_kernel_stats.csv
rm -rf $O/prof
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/s65/istft4096_kernel_stats.csv')))
for r in rows[:6]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(4), '%.3f'%(float(r['AverageNs'])/1e6), r['Percentage'])
PY
