#!/bin/bash
# round 3, session 38: the corrected LUFS workspace test, the tap-design and tiled tests once more on the final binary
cd $GRAFT_REPO_ROOT
O=gpurun_out/s77; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lufs or loud or tap_design or istft_tiled" 2>&1 | tail -4 | tee $O/pytest.log
