#!/bin/bash
# round 4, session 21: long FIRs (HighPass's default cutoffs) as one four-step convolution
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s21; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -k "fir or sinc or pass or filter or equal or gradient" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -8 | cut -c1-300
timeout 300 python tools/tfmbench.py 2>&1 | grep -v amdgpu.ids | grep -e "^batch" -e Pass -e Equalizer | tee $O/tfmbench.log
AT_FIR_METHOD=fft timeout 300 python tools/tfmbench.py 2>&1 | grep -v amdgpu.ids | grep -e HighPass | sed 's/^/AT_FIR_METHOD=fft  /' | tee -a $O/tfmbench.log
