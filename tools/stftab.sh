#!/bin/bash
export AT_DEV_KNOBS=1   # A/B script: development build of the library
# A/B of the STFT kernels on one box: {scalar, SLP-packed} build x {generic, v2} kernel.
# usage: tools/stftab.sh [batch]   (run on the GPU box; prints kbench lines)
B=${1:-512}
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
for lib in libaudiotools_amd.so libaudiotools_amd_slp.so; do
  [ -f $R/audiotools_amd/lib/$lib ] || continue
  for v2 in 0 1; do
    echo "### lib=$lib AT_STFT_V2=$v2 batch=$B"
    AT_LIB_PATH=$R/audiotools_amd/lib/$lib AT_STFT_V2=$v2 python $R/tools/kbench.py --what stft,stftmel --iters 30 --batch $B
  done
done
