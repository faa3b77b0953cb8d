#!/bin/bash
# round 3, session 42: istft.hip under the max-ilp scheduler vs the default one, all inverse-kernel users
cd $GRAFT_REPO_ROOT
O=gpurun_out/s82; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
( timeout 500 python -m pytest tests -m gpu -q -x -k "istft or inverse or edit or grad or adjoint or spectral or roundtrip or round_trip" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
for lib in libaudiotools_amd_base.so libaudiotools_amd.so; do
  echo "### rep $rep lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what istft --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what istft --iters 20 --sr 22050 --nfft 1024 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what istft --iters 20 --sr 16000 --nfft 512 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what istft --iters 50 --batch 64 2>&1 | grep -v -e amdgpu.ids -e "^$"
done; done 2>&1 | tee $O/ab.log
for lib in libaudiotools_amd_base.so libaudiotools_amd.so; do
  echo "### lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/specbench.py 256 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/gradbench.py 128 2>&1 | grep -v -e amdgpu.ids -e "^$"
done 2>&1 | tee $O/ab2.log
