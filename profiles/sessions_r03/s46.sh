#!/bin/bash
# round 3, session 7: v2 STFT with last-pass twiddle powers from LDS (flags bit 6) A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/s46; mkdir -p $O
( timeout 150 python tools/stftsweep.py --batch 512 --mel 1 --iters 20 --reps 7 --cfg 72:0:1,72:0:65,72:0:0,72:0:64 ;
  timeout 60 python tools/stftsweep.py --batch 64 --mel 1 --iters 20 --reps 7 --cfg 72:0:1,72:0:65 ) > $O/stft.log 2>&1
grep -v amdgpu $O/stft.log
