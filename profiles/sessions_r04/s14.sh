#!/bin/bash
# round 4, session 14: one-pass inverse of the run-time sizes (overlap-add in LDS): parity, then A/B against the frame buffer
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s14; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -q -k "generic or istft" 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
for cfg in "16000 400" "24000 1200" "48000 1920" "16000 800" "16000 320"; do
  set -- $cfg
  for mode in "AT_ISTFT_GENERIC_OLA=0" "AT_ISTFT_GENERIC_OLA=1" "AT_ISTFT_GENERIC_WGS=2"; do
    echo "### sr=$1 n_fft=$2 $mode" | tee -a $O/kbench.log
    env $mode timeout 200 python tools/kbench.py --what istft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  done
done
