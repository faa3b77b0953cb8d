#!/bin/bash
# round 3, session 28: tiled inverse tests again (short-row case fixed), spectral round trip at 96 kHz
cd $GRAFT_REPO_ROOT
O=gpurun_out/s67; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "istft_tiled or generic_match" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
