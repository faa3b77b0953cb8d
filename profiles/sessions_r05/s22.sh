#!/bin/bash
# round 5, session 22: the inverse STFT with its signal buffer from the placement pool: semantics test, inverse tests, timing
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s22; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "placement or istft or inverse or round_trip or Spectral" 2>&1 | tail -4 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do timeout 300 python tools/kbench.py --what istft,stft --iters 30 2>&1 | grep -v Warn | tail -5; done > $O/kbench.log 2>&1; cat $O/kbench.log
