#!/bin/bash
# round 5, session 34: shipped build with the non-temporal 512-byte runs: transform / mel / golden tests; then n_fft 1024 + mel
# (old store path, 256-byte segments) with non-temporal segments, development builds A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s34; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so | cut -c1-16
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_oracle_golden.py tests/test_golden_r05.py -m gpu -q -k "stft or mel or mfcc or golden or transform or Spectral or round_trip" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
export AT_DEV_KNOBS=1
for round in 1 2; do
for lib in libaudiotools_amd_dev libat_segnt; do
  echo "### $lib round $round"
  for cfg in "1024 44100" "1024 22050"; do set -- $cfg
    echo "# n_fft $1 @ $2"
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib.so timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
  done
done
done > $O/ab.log 2>&1
cat $O/ab.log
