#!/bin/bash
# round 6, session 4: the north-star line with the new bench flow (pool opted in for the timed region, plain pass after it),
# then the round's profile evidence (kernel stats + PMC passes) on the same box
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s04; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; p=r["placement"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "kernel ms", r["avg_launch_ms"], "floor same", r["floor_ms_same_buffers"])
print("plain: kernel", p.get("kernel_ms_plain_allocation"), "frac", p.get("frac_plain_allocation"), "ms/step", p.get("ms_per_step_plain_allocation"))
print("pool", p.get("pool"))
print("stft_only", r["stft_only"]); print("lufs", d["kernels_ms"]); print("share", d.get("share_64")); print("parity", d["parity_check"]["ok"])
PY
bash tools/profile_round.sh r06 > $O/profile.log 2>&1; tail -c 1500 $O/profile.log
cp gpurun_out/profile_r06/summary.json $O/pmc_summary.json; cp gpurun_out/profile_r06/kernel_stats.csv $O/kernel_stats.csv
