#!/bin/bash
# round 4, session 8: compile-time pass lists for the speech windows (n_fft 400 / 1200 / 1920): parity and A/B against the run-time plan
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s08; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "generic or many_frames" 2>&1 | tail -8 ) > $O/pytest_generic.log 2>&1
tail -3 $O/pytest_generic.log
for cfg in "16000 400" "24000 1200" "48000 1920"; do
  set -- $cfg
  for pl in 0 1; do
    echo "### sr=$1 n_fft=$2 AT_STFT_GENERIC_PLANS=$pl" | tee -a $O/kbench_generic.log
    AT_STFT_GENERIC_PLANS=$pl timeout 200 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench_generic.log
  done
done
