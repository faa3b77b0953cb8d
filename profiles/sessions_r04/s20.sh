#!/bin/bash
# round 4, session 20: every loader-free transform on one HIP batch -- looking for slow torch fallbacks
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s20; mkdir -p $O
timeout 600 python tools/tfmbench.py 2>&1 | grep -v amdgpu.ids | tee $O/tfmbench.log
