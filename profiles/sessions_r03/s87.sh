#!/bin/bash
# round 3, session 47: four scheduler options on stft.hip (RP trackers, no post-RA scheduler, no re-schedule stages, no memop clustering)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s87; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
for rep in 1 2 3; do
for v in "" _v_trk _v_nopost _v_norr _v_noclu; do
  echo "### rep $rep lib=libaudiotools_amd$v.so"
  AT_LIB_PATH=$L/libaudiotools_amd$v.so timeout 200 python tools/kbench.py --what stft,stftmel --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$"
done; done 2>&1 | tee $O/ab.log
