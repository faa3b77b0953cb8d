#!/bin/bash
# round 5, session 4: span schedules of the v2 kernel and its twin on 5 buffer sets (dev build): every XCD span starting its
# rounds at a different phase; row-interleaved spans
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s04; mkdir -p $O
export AT_DEV_KNOBS=1 AT_STFT_TUNE=1
timeout 300 python tools/regime.py --sets 5 --spans --tag spans > $O/regime_spans.log 2>&1
tail -n 70 $O/regime_spans.log
