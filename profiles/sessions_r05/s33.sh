#!/bin/bash
# round 5, session 33: the 512-byte runs of the several-frames-per-wave kernels as non-temporal stores (development builds, A/B)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5s33; mkdir -p $O
for round in 1 2; do
for lib in libaudiotools_amd_dev libat_nt; do
  echo "### $lib round $round"
  for cfg in "512 16000" "1024 44100" "256 8000"; do set -- $cfg
    echo "# n_fft $1 @ $2"
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib.so timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
  done
done
done > $O/ab.log 2>&1
cat $O/ab.log
