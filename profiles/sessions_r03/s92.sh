#!/bin/bash
# round 3, session 52: inverse STFT window table as per-lane rows (16-byte reads) vs the s91 binary
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s92; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
( timeout 300 python -m pytest tests -m gpu -q -x -k "istft or inverse or edit or spectral or roundtrip or round_trip or golden or grad or adjoint or loss or vocoder or pitch or stretch or gate" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
for lib in libaudiotools_amd_base.so libaudiotools_amd.so; do
  echo "### rep $rep lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 100 python tools/kbench.py --what istft --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 100 python tools/kbench.py --what istft --iters 50 --batch 64 2>&1 | grep -v -e amdgpu.ids -e "^$"
done; done 2>&1 | tee $O/ab.log
