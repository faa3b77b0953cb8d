#!/bin/bash
# round 4, session 4: resampler depth A/B, the two-stream question, the FIR meter on the HIP path,
# counters of the north-star step, of cfg4, and of the inverse / tiled transforms at full size
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s04; mkdir -p $O
( timeout 600 python -m pytest tests/test_golden_r04.py tests/test_gpu_parity.py -m gpu -q -k "fir_meter or resample_f16 or resample_structured or lufs or loudness" 2>&1 | tail -8 ) > $O/pytest_sub.log 2>&1
tail -3 $O/pytest_sub.log
timeout 300 python tools/rsbench.py --iters 20 --rounds 3 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
timeout 300 python tools/twostream.py 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/twostream.log
bash tools/profile_round.sh r04_bench > $O/profile_bench.log 2>&1
bash tools/profile_round.sh r04_cfg4 --config cfg4 > $O/profile_cfg4.log 2>&1
export PMC_FILTER="istft|tiled|stft_mel"
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh $O/pmc_istft_$c $c -- python tools/kbench.py --what istft,stft --iters 3 2>&1 | grep -v amdgpu.ids | tee -a $O/pmc_fullsize.log
  bash tools/pmc.sh $O/pmc_4096_$c $c -- python tools/kbench.py --what stft,genmel,istft --iters 3 --sr 96000 --nfft 4096 --batch 256 2>&1 | grep -v amdgpu.ids | tee -a $O/pmc_fullsize.log
done
python - <<'PY'
import json
for tag in ("r04_bench", "r04_cfg4"):
    try:
        d = json.load(open(f"gpurun_out/profile_{tag}/summary.json"))
        print(tag, d.get("box"))
        for k, v in d["pmc_fetch"].items():
            w = d["pmc_write"].get(k, {}).get("WRITE_SIZE")
            print("  ", k[:70], "FETCHx2+WRITE GB:", None if w is None else round((2 * v["FETCH_SIZE"] + w) * 1024 / 1e9, 3))
        for r in d["kernel_stats"][:9]:
            print("  ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
    except Exception as e:
        print(tag, "failed", e)
PY
