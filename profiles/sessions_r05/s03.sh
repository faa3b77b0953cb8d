#!/bin/bash
# round 5, session 3: the schedule sweep of s01 repeated with the switch actually on (AT_STFT_TUNE must be in the environment
# BEFORE the library's first call): XCD spans x run length, loads-only / stores-only twins, on the slowest and fastest of 5 sets.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s03; mkdir -p $O
export AT_DEV_KNOBS=1 AT_STFT_TUNE=1
timeout 300 python tools/regime.py --sets 5 --variants --sched --tag sched > $O/regime_sched.log 2>&1
tail -n 80 $O/regime_sched.log
