#!/bin/bash
# round 4, session 11: many inverse frames per tile for the speech windows (istft_frames_generic_tiled_kernel) -- parity,
# then old / new A/B on 512 rows x 10 s; the north-star line with the floor twin timed both ways
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s11; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -q -k "generic or istft" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for cfg in "16000 400" "24000 1200" "48000 1920" "44100 882" "16000 800"; do
  set -- $cfg
  for mode in "AT_ISTFT_GENERIC_OLD=1" "AT_ISTFT_GENERIC_OLD=0" "AT_ISTFT_GENERIC_WGS=3" "AT_ISTFT_GENERIC_WGS=2" "AT_STFT_GENERIC_PLANS=0"; do
    echo "### sr=$1 n_fft=$2 $mode" | tee -a $O/kbench.log
    env $mode timeout 200 python tools/kbench.py --what istft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  done
done
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-2600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o istft400 -- python $GRAFT_REPO_ROOT/tools/kbench.py --what istft --iters 10 --batch 256 --sr 16000 --nfft 400 > $O/prof_istft400.log 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/s11/prof/**/*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:6]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
