#!/bin/bash
# round 5, session 35: FINAL binary (non-temporal 512-byte runs; non-temporal segments for n_fft 1024 + mel): transform tests,
# bench lines, rocprofv3 --kernel-trace --stats + one --pmc pass per counter group for the three configurations
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s35; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so > $O/lib_sha256.txt; cut -c1-16 $O/lib_sha256.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_oracle_golden.py -m gpu -q -k "stft or mel or mfcc or golden or placement or north_star" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; tail -1 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_ns.log 2>&1
timeout 900 bash tools/profile_round.sh r05_bench > $O/profile_bench.log 2>&1
grep '^{' gpurun_out/profile_r05_bench/bench_stats.log | tail -1 > $O/bench_stats_line.json
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1
timeout 900 bash tools/profile_round.sh r05_cfg4 --config cfg4 > $O/profile_cfg4.log 2>&1
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1
timeout 1200 bash tools/profile_round.sh r05_cfg5 --config cfg5 > $O/profile_cfg5.log 2>&1
for t in bench cfg4 cfg5; do cp gpurun_out/profile_r05_$t/summary.json $O/r05_${t}_pmc_summary.json; cp gpurun_out/profile_r05_$t/kernel_stats.csv $O/r05_${t}_kernel_stats.csv; done
rm -rf gpurun_out/profile_r05_bench/pmc_* gpurun_out/profile_r05_cfg4/pmc_* gpurun_out/profile_r05_cfg5/pmc_* gpurun_out/profile_r05_*/stats
python3 - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_ns.log","bench_stats_line.json","bench_cfg4.log","bench_cfg5.log"):
    d=json.loads(open(f"{O}/{f}").read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
    print(f, "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "frac", round(r["frac"],4), "kernel ms", round(r["avg_launch_ms"],4), "twin own", r.get("floor_ms_same_buffers"), "plain", p.get("kernel_ms_plain_allocation"), (d.get("share_64") or {}).get("ms_per_step"), d.get("kernels_ms"))
for t in ("bench","cfg4","cfg5"):
    s=json.load(open(f"{O}/r05_{t}_pmc_summary.json"))
    for k in s["kernel_stats"][:4]: print(t, k["Name"][:60], k["Calls"], round(float(k["AverageNs"])/1e3,1))
PY
