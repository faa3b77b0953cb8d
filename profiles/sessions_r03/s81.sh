#!/bin/bash
# round 3, session 41: A/B of the scheduler strategy (-mllvm -amdgpu-sched-strategy=max-ilp) on the transform kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/s81; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
for rep in 1 2; do
for lib in libaudiotools_amd.so libaudiotools_amd_maxilp.so; do
  echo "### rep $rep lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/kbench.py --what stft,genmel,istft --iters 10 --batch 256 --sr 96000 --nfft 4096 2>&1 | grep -v -e amdgpu.ids -e "^$"
done; done 2>&1 | tee $O/ab.log
for lib in libaudiotools_amd.so libaudiotools_amd_maxilp.so; do
  echo "### lib=$lib"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/cfgbench.py --only applyir,chain 2>&1 | grep "cfg4 Room\|cfg4 full"
  AT_LIB_PATH=$L/$lib timeout 200 python tools/rsbench.py 2>&1 | grep -v -e amdgpu.ids -e "^$" | tail -4
done 2>&1 | tee $O/ab2.log
