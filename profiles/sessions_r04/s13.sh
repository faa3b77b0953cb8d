#!/bin/bash
# round 4, session 13: gradients through resample / convolve / apply_ir on the kernels
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s13; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x -k "gradient or grad" 2>&1 | tail -30 ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log
