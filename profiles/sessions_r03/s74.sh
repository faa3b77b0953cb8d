#!/bin/bash
# round 3, session 35: device-side filter design (one launch instead of ~25 / ~8 torch launches per call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s74; mkdir -p $O
( timeout 400 python -m pytest tests -m gpu -q -x -k "tap_design or low_pass or high_pass or equalizer or fir or sinc or cfg4 or golden or apply_ir or transform" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python tools/cfgbench.py --only lowpass,eq,applyir,chain > $O/cfg.log 2>&1; grep -v amdgpu $O/cfg.log | tail -7
timeout 200 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
