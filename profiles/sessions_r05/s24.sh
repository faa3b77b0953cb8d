#!/bin/bash
# round 5, session 24: 512-byte-run stores for the wave kernels with the mel stage too (magnitudes taken at read-back)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5s24; mkdir -p $O
( AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/libat_rs.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_oracle_golden.py -m gpu -q -k "stft or mel or mfcc or golden" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for round in 1 2; do
for lib in libaudiotools_amd_dev libat_rs; do
  echo "### $lib round $round"
  for cfg in "512 16000" "256 8000" "1024 22050" "1024 44100" "128 8000"; do set -- $cfg
    echo "# n_fft $1 @ $2"
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib.so timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
  done
done
done > $O/ab.log 2>&1
cat $O/ab.log
