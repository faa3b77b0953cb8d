#!/bin/bash
# round 5, session 16: kernels.stft_mel(out=...) + the placement probe of bench.py (the real kernel on the best of five
# placements of this process, next to the placement the API call got)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s16; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "caller_buffers or north_star_full" 2>&1 | tail -4 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ns_$i.log 2>&1; python3 - $O/bench_ns_$i.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("ms_per_step", round(d["ms_per_step"],3), "kernel", round(r["avg_launch_ms"],3), "frac", round(r["frac"],4), "twin own buffers", round(r["floor_ms_same_buffers"],3), "probe", {k:(([round(t,3) for t in v]) if isinstance(v,list) else (round(v,4) if isinstance(v,float) else None)) for k,v in (r["placement_probe"] or {}).items() if k!="note"})
PY
done
