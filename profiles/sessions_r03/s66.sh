#!/bin/bash
# round 3, session 27: tiled inverse transform for n_fft 4096 / 8192 (A/B against the frame buffer + gather path)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s66; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "istft_tiled or generic or istft" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for off in 1 0; do
  echo "### AT_ISTFT_TILED_OFF=$off  96 kHz n_fft 4096 B=256x2x10s; 192 kHz 8192 B=128"
  AT_ISTFT_TILED_OFF=$off timeout 120 python tools/kbench.py --what istft --iters 10 --batch 256 --sr 96000 --nfft 4096 2>&1 | grep istft
  AT_ISTFT_TILED_OFF=$off timeout 120 python tools/kbench.py --what istft --iters 10 --batch 128 --sr 192000 --nfft 8192 2>&1 | grep istft
done > $O/istft.log 2>&1
cat $O/istft.log
