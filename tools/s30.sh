#!/bin/bash
# round 2, GPU session 30: sanity tests + schedule / cache-policy / occupancy sweep of the v2 STFT kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/s30; mkdir -p $O
L=audiotools_amd/lib
( timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest.log 2>&1
S="timeout 120 python tools/stftsweep.py"
{
$S --runs 8,12,14,15,16,17,18,20,24,27,28,31,32,33,36,40,48,54,64,72,108,144,216,431
$S --runs 16,54,431 --nx 1,2,4,16
$S --runs 16,54 --flags 1,2,4
$S --mel 0 --runs 16,27,54,431 --flags 0,2
$S --nw 12 --runs 16,24,36,48,96,288 --tag nw12
$S --nw 12 --mel 0 --runs 16,24,36,48,96,288 --tag nw12
$S --batch 64 --iters 20 --runs 7,9,14,16,18,27,54
$S --batch 64 --iters 20 --nw 12 --runs 6,9,12,18,36
for v in prio1 prio2 stag4 stag16; do
  AT_LIB_PATH=$PWD/$L/libaudiotools_amd_$v.so $S --runs 16,54 --tag $v
  AT_LIB_PATH=$PWD/$L/libaudiotools_amd_$v.so $S --mel 0 --runs 16,54 --tag $v
done
} > $O/sweep.log 2>&1
{
for u in 2048 4096 8192 16384 32768; do
  echo "### AT_ISTFT_UNITS=$u"; AT_ISTFT_UNITS=$u timeout 120 python tools/kbench.py --what istft --iters 20 2>&1 | grep -v amdgpu.ids
done
timeout 120 python tools/kbench.py --what stft,stftmel,lufs,copy --iters 20 2>&1 | grep -v amdgpu.ids
} > $O/istft.log 2>&1
tail -3 $O/pytest.log; grep -c True $O/sweep.log; grep -c False $O/sweep.log
