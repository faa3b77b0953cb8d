#!/bin/bash
# round 4, session 32: non-temporal spectrum stores in the n_fft <= 1024 wave kernels (-DAT_STFT_OLD_NT=1), interleaved
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s32; mkdir -p $O
for rep in 1 2; do
for v in base nt; do
  if [ $v = base ]; then unset AT_LIB_PATH; else export AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_$v/libat.so; fi
  for cfg in "16000 512" "22050 1024" "8000 256" "44100 1024"; do
    set -- $cfg
    echo "## $v sr=$1 n_fft=$2" | tee -a $O/kbench.log
    timeout 200 python tools/kbench.py --what stft,stftmel --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  done
done
done
