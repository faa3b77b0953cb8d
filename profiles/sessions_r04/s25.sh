#!/bin/bash
# round 4, session 25: the wave kernels of n_fft 512 / 1024 / 256 split into "everything but the stores" and "stores only"
# (a -DAT_STFT_DEBUGMODES=1 build; AT_STFT_DEBUG=1 no stores, =2 stores only, same addresses)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s25; mkdir -p $O
export AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_d/libat.so
for cfg in "16000 512" "22050 1024" "8000 256" "44100 2048"; do
  set -- $cfg
  for dbg in 0 1 2; do
    echo "### sr=$1 n_fft=$2 AT_STFT_DEBUG=$dbg" | tee -a $O/kbench.log
    AT_STFT_DEBUG=$dbg AT_STFT_V2=0 timeout 200 python tools/kbench.py --what stft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  done
done
