#!/bin/bash
# round 3, session 24: cache-policy flags of the v2 forward kernel on this box (static-store variant), interleaved twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/s63; mkdir -p $O
for rep in 1 2 3; do
for f in 1 0 3 2 5 4; do
  echo "### AT_STFT_FLAGS=$f"
  AT_STFT_FLAGS=$f timeout 60 python tools/kbench.py --what stftmel --iters 30 2>&1 | grep "stft+mel"
done
done > $O/flags.log 2>&1
cat $O/flags.log
timeout 60 python tools/kbench.py --what copy,fill --iters 10 2>&1 | grep -v amdgpu
