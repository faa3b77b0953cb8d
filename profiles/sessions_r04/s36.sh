#!/bin/bash
# round 4, session 36: the rewritten long-FIR test
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s36; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fir_long" 2>&1 | tail -3 | tee $O/pytest.log
