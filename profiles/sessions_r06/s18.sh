#!/bin/bash
# round 6, session 18: stft.hip built with the max-ilp scheduling strategy (A/B build; the mel kernel takes 256 registers + 4 scratch dwords with it)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
for round in 1 2; do
for lib in libaudiotools_amd_dev libat_ilp; do
  echo "### $lib round $round"
  AT_LIB_PATH=$L/$lib.so timeout 200 python tools/kbench.py --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
done
done
