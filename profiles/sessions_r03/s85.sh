#!/bin/bash
# round 3, session 45: resident waves per SIMD of the small-size forward kernels (n_fft 128 ... 1024): 2 (shipped) vs 3, 4
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s85; mkdir -p $O
for rep in 1 2; do
for wps in 2 3 4; do
  for cfg in "8000 256" "16000 512" "22050 1024" "8000 128"; do
    set -- $cfg
    echo "### rep $rep wps=$wps sr=$1 n_fft=$2"
    AT_STFT_V1_WPS=$wps timeout 200 python tools/kbench.py --what stft,stftmel --iters 20 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$"
  done
done; done 2>&1 | tee $O/ab.log
