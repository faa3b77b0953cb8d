#!/bin/bash
# round 4, session 34: parity of the generic / tiled transforms on the build with non-temporal stores in the forward tile
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s34; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -k "generic or tiled or 4096 or 8192 or stride" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -4 | cut -c1-300
