#!/bin/bash
# round 6, session 9: the inverse transform at n_fft 64 / 128 / 256 through the generic one-pass kernel (consecutive frames per tile,
# overlap-add in LDS) instead of the fused one (frame slots on different row segments: 32-byte load pieces at 128) -- development A/B
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
for nfft in 64 128 256 512; do
  for knob in 0 512; do
    echo "### n_fft $nfft AT_ISTFT_SMALL_OLA=$knob"
    AT_ISTFT_SMALL_OLA=$knob timeout 200 python tools/kbench.py --nfft $nfft --sr 8000 --what istft --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
  done
done
AT_ISTFT_SMALL_OLA=512 python - <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from audiotools_amd import kernels, tables
x = (0.1 * torch.randn(8, 2, 80000, device='cuda')).clamp_(-1, 1)
for n_fft in (64, 128, 256, 512):
    hop = n_fft // 4
    win = tables.window('hann', n_fft, x.device)
    X, _ = kernels.stft_mel(x, win, n_fft, hop)
    y = kernels.istft(X, win, n_fft, hop, x.shape[-1])
    ref = torch.istft(X.reshape(16, n_fft // 2 + 1, -1), n_fft, hop, window=win, length=x.shape[-1]).reshape(x.shape)
    print(n_fft, "generic-ola vs torch.istft", float((y - ref).abs().max() / ref.abs().max()), "round trip", float((y - x).abs().max()))
PY
