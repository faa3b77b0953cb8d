#!/bin/bash
# round 3, session 25: wave kernels for n_fft <= 1024 with the next group's samples requested one group ahead (A/B, same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s64; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stft and not generic and not structured" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
for rep in 1 2; do
for v in alt_old.so libaudiotools_amd.so; do
  for cfg in "16000 512" "22050 1024" "8000 256" "44100 1024"; do
    set -- $cfg
    echo "### $v sr=$1 n_fft=$2"
    AT_LIB_PATH=$L/$v timeout 100 python tools/kbench.py --what stft,stftmel --iters 20 --batch 512 --sr $1 --nfft $2 2>&1 | grep "stft"
  done
done
done > $O/small.log 2>&1
cat $O/small.log
