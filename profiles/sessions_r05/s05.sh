#!/bin/bash
# round 5, session 5: is there an affinity between an XCD and a physical region of the spectrum buffer?  One XCD's workgroups
# at a time (the others leave at once) on each eighth of the slowest and the fastest of 5 buffer sets.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s05; mkdir -p $O
export AT_DEV_KNOBS=1 AT_STFT_TUNE=1
timeout 300 python tools/regime.py --sets 5 --affinity --tag affinity > $O/regime_affinity.log 2>&1
tail -n 40 $O/regime_affinity.log
