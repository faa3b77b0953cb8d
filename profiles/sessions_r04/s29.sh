#!/bin/bash
# round 4, session 29: start-up stagger of the wave slots of a CU for the n_fft <= 1024 wave kernels (s28: their compute
# chain and their HBM write stream add up instead of overlapping -- are the waves of the chip in lockstep?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s29; mkdir -p $O
for rep in 1 2; do
for v in base s8 s27 s80; do
  if [ $v = base ]; then unset AT_LIB_PATH; else export AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_$v/libat.so; fi
  for cfg in "16000 512" "22050 1024" "8000 256"; do
    set -- $cfg
    echo -n "$v sr=$1 n_fft=$2  " | tee -a $O/kbench.log
    timeout 200 python tools/kbench.py --what stft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  done
done
done
