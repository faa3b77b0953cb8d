#!/bin/bash
# round 5, session 14: alter_drr with the row in registers (one read instead of three): parity, then cfg4 per kernel
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s14; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden_r03.py tests/test_golden_r05.py -m gpu -q -k "alter_drr or ir_tools or apply_ir or golden or RoomImpulse or threshold" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for i in 1 2; do timeout 200 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 ms_per_step', round(d['ms_per_step'],3), d['parity_check']['chain_rel'])"; done | tee $O/cfg4.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --steps 40 --warmup 10 --no-cpu-baseline > $O/kt.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/kt/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    print(r["Name"][:84].ljust(84), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
