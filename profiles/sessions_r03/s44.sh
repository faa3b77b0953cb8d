#!/bin/bash
# round 3, session 5: fused STFT-domain edits in the inverse kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/s44; mkdir -p $O
( timeout 300 python -m pytest tests -m gpu -q -x -k "deferred or fused_inverse or spectral or istft or edits or transforms_gpu" 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 120 python tools/specbench.py > $O/spec.log 2>&1; grep -v amdgpu $O/spec.log | tail -20
