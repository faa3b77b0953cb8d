#!/bin/bash
# round 3, session 19: colfft with register-staged loads (all of a thread's segments in flight); cfg4 stages
cd $GRAFT_REPO_ROOT
O=gpurun_out/s58; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fourstep or longconv or convol or apply_ir or room or cfg4" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python tools/cfgbench.py --only lowpass,eq,applyir,chain > $O/cfg.log 2>&1; grep -v amdgpu $O/cfg.log | tail -8
timeout 100 python tools/convbench.py > $O/conv.log 2>&1; grep -v amdgpu $O/conv.log | tail -12
