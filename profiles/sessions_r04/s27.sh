#!/bin/bash
# round 4, session 27: corner shapes of the one-pass inverse
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s27; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -k "one_pass_edge or istft" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -25 | cut -c1-400
