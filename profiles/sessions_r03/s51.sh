#!/bin/bash
# round 3, session 12: generic STFT, hand-addressed tile for the power-of-two plans (A/B against the generic tile, same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s51; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "generic or 4096 or 96" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for rep in 1 2; do
for old in 1 0; do
  echo "### AT_STFT_TILED_OLD=$old  96 kHz n_fft 4096 B=256x2x10s"
  AT_STFT_TILED_OLD=$old timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr 96000 --nfft 4096
  echo "### AT_STFT_TILED_OLD=$old  192 kHz n_fft 8192 B=128x2x10s"
  AT_STFT_TILED_OLD=$old timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 128 --sr 192000 --nfft 8192
done
done > $O/generic.log 2>&1
grep -v amdgpu $O/generic.log
