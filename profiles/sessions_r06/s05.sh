#!/bin/bash
# round 6, session 5: padding units of the mel tables read their predecessor's row (bank-conflict-free unit reads): the mel /
# stft GPU tests, then the north-star line and the round's profile evidence with this binary
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel or stft or north_star or cfg2 or cfg5" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do python tools/kbench.py --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids; done
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; p=r["placement"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "kernel ms", r["avg_launch_ms"], "floor same", r["floor_ms_same_buffers"])
print("plain: kernel", p.get("kernel_ms_plain_allocation"), "frac", p.get("frac_plain_allocation"), "ms/step", p.get("ms_per_step_plain_allocation"))
print("stft_only", r["stft_only"]); print("lufs", d["kernels_ms"]); print("share", d.get("share_64")); print("parity", d["parity_check"]["ok"])
PY
bash tools/profile_round.sh r06 > $O/profile.log 2>&1
cp gpurun_out/profile_r06/summary.json $O/pmc_summary.json; cp gpurun_out/profile_r06/kernel_stats.csv $O/kernel_stats.csv
python - <<'PY'
import json
d=json.load(open('gpurun_out/profile_r06/summary.json'))
F=882688
for k in d['kernel_stats'][:2]: print(k['Name'][:70], k['Calls'], float(k['AverageNs'])/1e6)
for grp in ('pmc_sq','pmc_lds'):
    for k,v in d[grp].items():
        if 'v2<4' in k: print(grp, {c: round(x/F,1) for c,x in v.items()})
PY
