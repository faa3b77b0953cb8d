#!/bin/bash
# round 5, session 13: does the regime depend on the SPACING of the eight XCD spans?  The twin on the first R rows of each of
# five buffer sets (spans of R / 8 rows): microseconds per row against R.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s13; mkdir -p $O
export AT_DEV_KNOBS=1
timeout 300 python tools/regime.py --sets 5 --spacing --tag spacing > $O/regime_spacing.log 2>&1
tail -n 12 $O/regime_spacing.log
