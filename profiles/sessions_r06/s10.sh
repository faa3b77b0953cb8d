#!/bin/bash
# round 6, session 10: n_fft <= 128 inverse through the generic one-pass kernel (shipped default): the inverse / transform tests, kbench rows
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -k "istft or roundtrip or spectral or transform or golden or stretch" 2>&1 | grep -v "^Extension modules" | tail -4
for nfft in 64 128 256; do echo "# n_fft $nfft @ 8 kHz"; timeout 200 python tools/kbench.py --nfft $nfft --sr 8000 --what stft,istft --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"; done
