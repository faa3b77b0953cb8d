#!/bin/bash
# round 4, session 9: speech-window plans at two vs three workgroups per CU (168-register build: 4-6 spilled registers)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s09; mkdir -p $O
for rep in 1 2; do
for cfg in "16000 400" "24000 1200" "48000 1920"; do
  set -- $cfg
  for lib in libaudiotools_amd.so alt_wgs3.so; do
    echo "### rep $rep sr=$1 n_fft=$2 $lib" | tee -a $O/kbench_wgs.log
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib timeout 200 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench_wgs.log
  done
done
done
