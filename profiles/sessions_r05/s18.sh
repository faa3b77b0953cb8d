#!/bin/bash
# round 5, session 18: placement-aware output pool ON (product default) -- the whole GPU suite + smoke, the three bench lines,
# rocprofv3 stats + counter passes of the north star on the same (final) binary
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s18; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so > $O/lib_sha256.txt
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_ns.log 2>&1
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1
timeout 900 bash tools/profile_round.sh r05_bench > $O/profile_bench.log 2>&1
cp gpurun_out/profile_r05_bench/summary.json $O/r05_bench_pmc_summary.json; cp gpurun_out/profile_r05_bench/kernel_stats.csv $O/r05_bench_kernel_stats.csv
rm -rf gpurun_out/profile_r05_bench/pmc_* gpurun_out/profile_r05_bench/stats
python3 - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_ns","bench_cfg4","bench_cfg5"):
    d=json.loads(open(f"{O}/{f}.log").read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
    print(f, "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "frac", round(r["frac"],4), "traffic", r.get("traffic"), "twin own", r.get("floor_ms_same_buffers"), "plain", p.get("kernel_ms_plain_allocation"), (d.get("share_64") or {}).get("ms_per_step"))
s=json.load(open(f"{O}/r05_bench_pmc_summary.json"))
for k in s["kernel_stats"][:4]: print(k["Name"][:60], k["Calls"], round(float(k["AverageNs"])/1e3,1))
PY
