#!/bin/bash
# round 3, session 15: generic STFT pow2 tile, mel epilogue riding on the next tile's barriers
cd $GRAFT_REPO_ROOT
O=gpurun_out/s54; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "generic or 4096 or 96 or mel" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for rep in 1 2; do
  echo "### 96 kHz n_fft 4096 B=256x2x10s"
  timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr 96000 --nfft 4096
  echo "### 192 kHz n_fft 8192 B=128x2x10s"
  timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 128 --sr 192000 --nfft 8192
done > $O/generic.log 2>&1
grep -v amdgpu $O/generic.log
