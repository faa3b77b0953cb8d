#!/bin/bash
# round 5, session 36: the rest of the GPU suite on the final binary ed317903 (everything s35's selection left out)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s36; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so | cut -c1-16
timeout 250 python -m pytest tests -m gpu -q -k "not (stft or mel or mfcc or golden or placement or north_star)" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
