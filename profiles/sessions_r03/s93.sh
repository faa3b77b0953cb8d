#!/bin/bash
# round 3, session 53: bench.py untraced vs under rocprofv3 --kernel-trace --stats, same box, last binary
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s93; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/box.txt
timeout 100 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-share 2>/dev/null | tail -1 > $O/untraced.json
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o b -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-share > $O/traced.log 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv; rm -rf $O/st
grep "^{" $O/traced.log | tail -1 > $O/traced.json
python3 - $O <<'P'
import json, sys, csv
O = sys.argv[1]
for n in ("untraced", "traced"):
    d = json.load(open(f"{O}/{n}.json")); print(n, "events avg_launch_ms", round(d["roofline"]["avg_launch_ms"], 4), "ms_per_step", round(d["ms_per_step"], 4))
for r in list(csv.DictReader(open(f"{O}/kernel_stats.csv")))[:2]:
    print("trace", r["Name"][:60], "calls", r["Calls"], "avg", round(float(r["AverageNs"]) / 1e6, 4), "min", round(float(r["MinNs"]) / 1e6, 4))
P
