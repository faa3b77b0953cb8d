#!/bin/bash
# round 6, session 17: run length of the v2 kernel now that a stretch drains its last frame's mel (PIPE): interleaved, bit-checked sweeps
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
python tools/stftsweep.py --batch 64 --iters 40 --reps 5 --cfg 72:0:1:8:1,72:0:1:8:2,72:0:1:8:3,14:0:1:8:1,11:0:1:8:1,9:0:1:8:1,72:0:1:16:3,72:0:1:4:3 2>&1 | grep -v Warn | grep -v amdgpu.ids
python tools/stftsweep.py --batch 512 --iters 10 --reps 5 --cfg 72:0:1:8:3,36:0:1:8:3,108:0:1:8:3,144:0:1:8:3,216:0:1:8:3,431:0:1:8:3 2>&1 | grep -v Warn | grep -v amdgpu.ids
