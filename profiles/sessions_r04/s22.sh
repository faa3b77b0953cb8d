#!/bin/bash
# round 4, session 22: fused quantization kernel, clip_distortion on the rows it reads; transforms again
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s22; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -k "quantiz or clip or transform or golden or fir_long" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -8 | cut -c1-300
timeout 300 python tools/tfmbench.py 2>&1 | grep -v amdgpu.ids | tee $O/tfmbench.log
