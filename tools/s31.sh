#!/bin/bash
# round 2, GPU session 31: interleaved A/B of the v2 STFT schedule knobs; iSTFT run planner
cd $GRAFT_REPO_ROOT
O=gpurun_out/s31; mkdir -p $O
S="timeout 150 python tools/stftsweep.py"
C1="16:0:0,431:0:0,72:0:0,54:4:0,54:16:0,431:2:0,431:4:0,431:8:0,431:16:0,431:24:0,431:32:0,431:48:0,72:16:0,216:16:0,431:16:8,431:0:8,431:16:1,431:16:4,431:16:5,431:0:16,431:16:16,431:16:13,16:16:0,16:0:0"
{
$S --cfg $C1
$S --mel 0 --cfg $C1
$S --batch 64 --iters 30 --reps 5 --cfg 16:0:0,18:0:0,54:0:0,54:2:0,54:4:0,54:8:0,54:16:0,27:4:0,27:16:0,54:8:8,54:16:8,18:8:0
$S --batch 256 --iters 15 --cfg 16:0:0,216:0:0,216:8:0,216:16:0,216:32:0,108:16:0
} > $O/sweep.log 2>&1
{
for b in 512 500 64 8; do
  echo "### batch $b (new planner)"; timeout 120 python tools/kbench.py --what istft --iters 20 --batch $b 2>&1 | grep -v amdgpu.ids
  echo "### batch $b AT_ISTFT_UNITS=16384 (old)"; AT_ISTFT_UNITS=16384 timeout 120 python tools/kbench.py --what istft --iters 20 --batch $b 2>&1 | grep -v amdgpu.ids
done
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "istft" 2>&1 | tail -3
} > $O/istft.log 2>&1
tail -4 $O/istft.log; grep -c True $O/sweep.log; grep -c False $O/sweep.log; true
