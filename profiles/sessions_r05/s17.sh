#!/bin/bash
# round 5, session 17: the placement-aware output pool -- semantics test, the tests around stft / autograd / transforms, two bench lines
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s17; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "placement or caller_buffers or north_star or stft or mel or autograd or grad" 2>&1 | tail -6 ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ns_$i.log 2>&1; python3 - $O/bench_ns_$i.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "kernel", round(r["avg_launch_ms"],3), "frac", round(r["frac"],4), "twin own", round(r["floor_ms_same_buffers"],3), "| plain:", round(p.get("kernel_ms_plain_allocation",0),3), round(p.get("frac_plain_allocation",0),4), "twin fresh", round(r["floor_ms"],3), "| pool", [( [round(t,3) for t in e["calibration_ms"]]) for e in p.get("pool",[])], p.get("error"))
PY
done
