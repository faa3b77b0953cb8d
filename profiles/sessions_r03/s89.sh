#!/bin/bash
# round 3, session 49: inverse STFT at n_fft 2048 with in-lane Hermitian pairs (plan 4.16.16, no ds_bpermute) vs the shipped kernel
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s89; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
( timeout 500 python -m pytest tests -m gpu -q -x -k "istft or inverse or edit or spectral or roundtrip or round_trip or golden" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2 3; do
for lib in libaudiotools_amd_base.so libaudiotools_amd.so; do
  echo "### rep $rep lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what istft --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what istft --iters 50 --batch 64 2>&1 | grep -v -e amdgpu.ids -e "^$"
done; done 2>&1 | tee $O/ab.log
for lib in libaudiotools_amd_base.so libaudiotools_amd.so; do
  echo "### lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/specbench.py 256 2>&1 | grep "folded\|istft alone"
done 2>&1 | tee $O/ab2.log
