#!/bin/bash
# round 5, session 11: cfg4 per kernel (is 5.08 ms of s10 a regression against r04's 4.90 or the box?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s11; mkdir -p $O
for i in 1 2; do timeout 200 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 ms_per_step', round(d['ms_per_step'],3), 'enqueue', round(d['config']['host_enqueue_ms'],3))"; done | tee $O/cfg4.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --steps 40 --warmup 10 --no-cpu-baseline > $O/kt.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/kt/**/*kernel_stats.csv",recursive=True)
tot=0
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:84].ljust(84), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
