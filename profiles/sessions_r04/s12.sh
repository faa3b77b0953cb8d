#!/bin/bash
# round 4, session 12: does the placement of the three buffers move the STFT + mel kernel or its zero-compute twin?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s12; mkdir -p $O
timeout 400 python tools/offsetbench.py --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/offsetbench.log
