#!/bin/bash
# round 6, session 8: closing evidence on one box with the binary of record: GPU suite + smoke, the three bench lines, the
# profile passes of the north star, per-kernel rows (kbench)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s08; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so | cut -c1-16
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^Extension modules" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
for cfg in north_star cfg4 cfg5; do
  python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"
done
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for cfg in ("north_star","cfg4","cfg5"):
    try:
        d=json.loads([l for l in open(f"{O}/bench_{cfg}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(cfg, "no line", e); continue
    r=d["roofline"]
    print(cfg, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "kernel ms", round(r["avg_launch_ms"],3), "parity", d.get("parity_check",{}).get("ok"), d.get("kernels_ms"))
    if cfg=="north_star":
        p=r["placement"]; print("   plain: kernel", p.get("kernel_ms_plain_allocation"), "frac", p.get("frac_plain_allocation"), "ms/step", p.get("ms_per_step_plain_allocation"), "floor same", r["floor_ms_same_buffers"], "stft_only", r["stft_only"]["avg_launch_ms"], "share", d["share_64"]["ms_per_step"], d["share_64"]["predicted_speedup_8gpu"])
PY
bash tools/profile_round.sh r06 > $O/profile.log 2>&1
cp gpurun_out/profile_r06/summary.json $O/pmc_summary.json; cp gpurun_out/profile_r06/kernel_stats.csv $O/kernel_stats.csv
python tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 2>&1 | grep -v Warn | grep -v amdgpu.ids
