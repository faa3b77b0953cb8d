#!/bin/bash
# round 3, session 39: alter_drr reports the peak the convolution needs (one pass over the impulse responses less)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s78; mkdir -p $O
( timeout 400 python -m pytest tests -m gpu -q -x -k "alter_drr or apply_ir or cfg4 or room or RoomImpulse or golden or ir_tools or drr or transform" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python tools/cfgbench.py --only applyir,chain 2>&1 | grep "cfg4 Room\|cfg4 full\|throughput" | tee $O/cfg.log
