#!/bin/bash
# round 5, session 21: placement survey -- which other kernels follow the physical placement of their input / output buffers
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s21; mkdir -p $O
timeout 900 python tools/placement_survey.py --k 5 --iters 8 > $O/survey.log 2>&1
grep -v Warn $O/survey.log | tail -20
