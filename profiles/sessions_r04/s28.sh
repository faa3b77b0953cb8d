#!/bin/bash
# round 4, session 28: the n_fft <= 1024 wave kernels with all stores redirected into a 2 MB region (AT_STFT_DEBUG=3):
# store ISSUE cost without the HBM write stream, next to 0 (normal), 1 (no stores)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s28; mkdir -p $O
export AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_d/libat.so
for cfg in "16000 512" "22050 1024" "8000 256"; do
  set -- $cfg
  for dbg in 0 1 3; do
    echo "### sr=$1 n_fft=$2 AT_STFT_DEBUG=$dbg" | tee -a $O/kbench.log
    AT_STFT_DEBUG=$dbg timeout 200 python tools/kbench.py --what stft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  done
done
cd /tmp && hipcc --offload-arch=gfx950 -O3 -o /tmp/segbench $GRAFT_REPO_ROOT/tools/micro/segbench.hip && /tmp/segbench 2>&1 | tail -14 | tee $O/segbench.log
