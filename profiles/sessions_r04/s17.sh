#!/bin/bash
# round 4, session 17: paired split step + scalar-base addressing in the run-time-size forward tile
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s17; mkdir -p $O
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -k "generic or mel" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -8 | cut -c1-300
for cfg in "16000 400" "24000 1200" "48000 1920" "16000 800" "44100 882"; do
  set -- $cfg
  echo "### sr=$1 n_fft=$2" | tee -a $O/kbench.log
  timeout 200 python tools/kbench.py --what stft,genmel,istft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
done
