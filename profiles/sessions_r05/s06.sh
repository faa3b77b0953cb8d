#!/bin/bash
# round 5, session 6: native adjoints of the run-time transform sizes and match_stride (new tests) + the transform / gradient tests around them
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s06; mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "autograd or adjoint or grad" 2>&1 | tail -15 ) > $O/pytest_adjoint.log 2>&1
( timeout 300 python -m pytest tests/test_golden_r05.py tests/test_golden_r04.py -m gpu -q -s 2>&1 | tail -12 ) > $O/pytest_golden.log 2>&1
cat $O/pytest_adjoint.log $O/pytest_golden.log
