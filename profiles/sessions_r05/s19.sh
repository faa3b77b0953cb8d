#!/bin/bash
# round 5, session 19: table staging of the v2 kernel with compile-time trip counts (all loads ahead of the LDS writes): parity,
# then the bench line with the new library and with the previous one (AT_LIB_PATH), alternating
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s19; mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_oracle_golden.py -m gpu -q -k "stft or mel or north_star or placement or golden" 2>&1 | tail -4 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
show() { python3 - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print(sys.argv[1].split("/")[-1], "ms_per_step", round(d["ms_per_step"],3), "kernel", round(r["avg_launch_ms"],4), "frac", round(r["frac"],4), "twin own", round(r["floor_ms_same_buffers"],3), "plain", round(p.get("kernel_ms_plain_allocation",0),3), "share", round((d.get("share_64") or {}).get("ms_per_step",0),4), d["lib_sha256"][:8])
PY
}
for i in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/new_$i.log 2>&1; show $O/new_$i.log
  AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/libaudiotools_amd_prev.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/prev_$i.log 2>&1; show $O/prev_$i.log
done
