#!/bin/bash
# round 3, session 16: generic pow2 tile, 2 vs 3 workgroups per CU; other generic sizes; cfg5 / rows refresh
cd $GRAFT_REPO_ROOT
O=gpurun_out/s55; mkdir -p $O
for rep in 1 2; do
for w in 2 3; do
  echo "### AT_STFT_TILED_WGS=$w  96 kHz n_fft 4096 B=256x2x10s"
  AT_STFT_TILED_WGS=$w timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr 96000 --nfft 4096
done
done > $O/generic.log 2>&1
echo "### 48 kHz n_fft 1920 B=256" >> $O/generic.log
timeout 120 python tools/kbench.py --what stft --iters 10 --batch 256 --sr 48000 --nfft 1920 >> $O/generic.log 2>&1
echo "### 16 kHz n_fft 512 B=512; 22.05 kHz n_fft 1024 B=512 (wave kernels)" >> $O/generic.log
timeout 120 python tools/kbench.py --what stft,stftmel --iters 10 --batch 512 --sr 16000 --nfft 512 >> $O/generic.log 2>&1
timeout 120 python tools/kbench.py --what stft,stftmel --iters 10 --batch 512 --sr 22050 --nfft 1024 >> $O/generic.log 2>&1
grep -v amdgpu $O/generic.log
