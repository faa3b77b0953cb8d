#!/bin/bash
# round 5, session 28: the inverse transform at the small sizes (how far from the forward's 50-55 % are they?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s28; mkdir -p $O
for cfg in "2048 44100" "1024 44100" "1024 22050" "512 16000" "256 8000" "128 8000"; do set -- $cfg
  echo "# n_fft $1 @ $2"; timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what istft --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool stft"
done > $O/istft.log 2>&1; cat $O/istft.log
