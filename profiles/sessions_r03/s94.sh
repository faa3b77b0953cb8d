#!/bin/bash
# round 3, session 54: the paired inverse layout at n_fft 1024 (M = 512: radices 4 . 16 . 8, composed last-pass twiddles)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s94; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
( timeout 100 python -m pytest tests -m gpu -q -x -k "istft or edit or roundtrip or round_trip" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for lib in libaudiotools_amd_base.so libaudiotools_amd.so libaudiotools_amd_base.so libaudiotools_amd.so; do
  echo "### lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 60 python tools/kbench.py --what istft --iters 20 --sr 22050 --nfft 1024 2>&1 | grep "istft"
  AT_LIB_PATH=$L/$lib timeout 60 python tools/kbench.py --what istft --iters 50 --sr 22050 --nfft 1024 --batch 64 2>&1 | grep "istft"
done 2>&1 | tee $O/ab.log
