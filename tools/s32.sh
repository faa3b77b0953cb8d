#!/bin/bash
# round 2, GPU session 32: lean mel rounds of the v2 kernel -- parity subset + interleaved A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/s32; mkdir -p $O
( timeout 250 python -m pytest tests -m gpu -x -q -k "mel or smoke or golden or cfg2" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
S="timeout 150 python tools/stftsweep.py"
{
$S --cfg 16:0:0,431:0:0,431:0:1,431:0:4,431:0:5,72:0:0,72:0:1,144:0:1,431:8:1,16:0:1,431:0:0,431:0:1
$S --batch 64 --iters 30 --reps 5 --cfg 16:0:0,18:0:0,54:0:0,18:0:1,54:0:1,54:0:5
$S --mel 0 --cfg 16:0:0,431:0:0,431:0:1
} > $O/sweep.log 2>&1
tail -3 $O/pytest.log; grep -v amdgpu $O/sweep.log
