#!/bin/bash
# round 3, session 34: schedule / cache-policy sweep of the v2 forward kernel with the static store count (interleaved, bit-checked)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s73; mkdir -p $O
timeout 300 python tools/stftsweep.py --iters 20 --reps 4 --cfg 72:0:1,72:0:0,144:0:1,36:0:1,431:0:1,72:4:1,72:8:1,72:16:1,72:0:9,72:0:17,72:0:3,72:0:5,54:0:1,108:0:1 2>&1 | grep -v amdgpu | tee $O/sweep.log
