#!/bin/bash
# round 5, session 20: the n_fft <= 512 wave kernels at three and four waves per SIMD (their 132-153 registers allow three
# without spills; four spills 40-130 B per lane), four or six waves per workgroup.  Development builds, same sources otherwise.
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5s20; mkdir -p $O
for round in 1 2; do
for lib in libaudiotools_amd_dev libat_wps3 libat_wps3nw6 libat_wps4; do
  echo "### $lib round $round"
  for cfg in "512 16000" "256 8000" "1024 22050"; do set -- $cfg
    echo "# n_fft $1 @ $2"
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib.so timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stft,stftmel --iters 30 2>&1 | grep -v Warn | tail -2
  done
done
done > $O/ab.log 2>&1
cat $O/ab.log
