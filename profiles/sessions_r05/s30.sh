#!/bin/bash
# round 5, session 30: the north star's rocprofv3 evidence again with the probes out of the --stats pass (s29's per-kernel
# average held 203 launches of the plain-allocation probe and the twin's probes next to the 211 timed ones)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s30; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so > $O/lib_sha256.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_ns.log 2>&1
timeout 900 bash tools/profile_round.sh r05_bench > $O/profile_bench.log 2>&1
cp gpurun_out/profile_r05_bench/summary.json $O/r05_bench_pmc_summary.json; cp gpurun_out/profile_r05_bench/kernel_stats.csv $O/r05_bench_kernel_stats.csv
tail -1 gpurun_out/profile_r05_bench/bench_stats.log > $O/bench_stats_line.json
rm -rf gpurun_out/profile_r05_bench/pmc_* gpurun_out/profile_r05_bench/stats
python3 - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_ns","bench_stats_line"):
    d=json.loads(open(f"{O}/{f}."+("log" if f=="bench_ns" else "json")).read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
    print(f, "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "frac", round(r["frac"],4), "kernel ms", round(r["avg_launch_ms"],4), "traffic", r.get("traffic"), "twin own", r.get("floor_ms_same_buffers"), "plain", p.get("kernel_ms_plain_allocation"), (d.get("share_64") or {}).get("ms_per_step"))
s=json.load(open(f"{O}/r05_bench_pmc_summary.json"))
for k in s["kernel_stats"][:5]: print(k["Name"][:60], k["Calls"], round(float(k["AverageNs"])/1e3,1))
PY
