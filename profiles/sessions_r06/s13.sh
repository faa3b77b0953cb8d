#!/bin/bash
# round 6, session 13: the north-star line of the binary of record on another box (box spread)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s13; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so | cut -c1-16
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; p=r["placement"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "kernel ms", round(r["avg_launch_ms"],3), "floor same", r["floor_ms_same_buffers"], "traffic", r["traffic"])
print("plain: kernel", p.get("kernel_ms_plain_allocation"), "frac", p.get("frac_plain_allocation"), "ms/step", p.get("ms_per_step_plain_allocation"))
print("stft per-step min/median/max", d["kernels_ms"].get("stft_mel_min"), d["kernels_ms"].get("stft_mel_median"), d["kernels_ms"].get("stft_mel_max")); print("stft_only", r["stft_only"]["avg_launch_ms"], "lufs", d["kernels_ms"]["lufs_total"], "share", d["share_64"]["ms_per_step"], d["share_64"]["predicted_speedup_8gpu"], "parity", d["parity_check"]["ok"], d["parity_check"]["every_rank_device_check"]["rank0_items_checked"])
PY
