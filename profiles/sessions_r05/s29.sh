#!/bin/bash
# round 5, session 29: evidence on the final binary (512-byte-run stores, pooled inverse, twelve candidates): bench lines,
# rocprofv3 --kernel-trace --stats + one --pmc pass per counter group for the three configurations
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s29; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so > $O/lib_sha256.txt
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "placement or north_star or into_caller" 2>&1 | tail -2 ) > $O/pytest.log 2>&1; tail -1 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_ns.log 2>&1
timeout 900 bash tools/profile_round.sh r05_bench > $O/profile_bench.log 2>&1
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1
timeout 900 bash tools/profile_round.sh r05_cfg4 --config cfg4 > $O/profile_cfg4.log 2>&1
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1
timeout 1200 bash tools/profile_round.sh r05_cfg5 --config cfg5 > $O/profile_cfg5.log 2>&1
for t in bench cfg4 cfg5; do cp gpurun_out/profile_r05_$t/summary.json $O/r05_${t}_pmc_summary.json; cp gpurun_out/profile_r05_$t/kernel_stats.csv $O/r05_${t}_kernel_stats.csv; done
rm -rf gpurun_out/profile_r05_bench/pmc_* gpurun_out/profile_r05_cfg4/pmc_* gpurun_out/profile_r05_cfg5/pmc_* gpurun_out/profile_r05_*/stats
python3 - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_ns","bench_cfg4","bench_cfg5"):
    d=json.loads(open(f"{O}/{f}.log").read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
    print(f, "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "frac", round(r["frac"],4), "kernel ms", round(r["avg_launch_ms"],4), "traffic", r.get("traffic"), "twin own", r.get("floor_ms_same_buffers"), "plain", p.get("kernel_ms_plain_allocation"), (d.get("share_64") or {}).get("ms_per_step"), d.get("kernels_ms"))
for t in ("bench","cfg4","cfg5"):
    s=json.load(open(f"{O}/r05_{t}_pmc_summary.json"))
    for k in s["kernel_stats"][:5]: print(t, k["Name"][:60], k["Calls"], round(float(k["AverageNs"])/1e3,1))
PY
