#!/bin/bash
# round 4, session 30: the n_fft <= 1024 wave kernels next group loaded BEFORE the transform (a group of lead time), every slot a real frame
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s30; mkdir -p $O
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -k "stft or mel or spec or golden or loss or mfcc or stride" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -6 | cut -c1-300
for cfg in "16000 512" "22050 1024" "8000 256" "16000 128" "44100 1024"; do
  set -- $cfg
  echo "### sr=$1 n_fft=$2" | tee -a $O/kbench.log
  timeout 200 python tools/kbench.py --what stft,stftmel,istft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
done
