#!/bin/bash
# round 3, session 2: full GPU suite with the per-row metric + new goldens/structured tests; cfg4 host side
cd $GRAFT_REPO_ROOT
O=gpurun_out/s41; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log
( timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "structured" 2>&1 | grep "per-row error" ) > $O/structured.log 2>&1
cat $O/structured.log | cut -c1-400
timeout 200 python tools/cfgbench.py --only chain,applyir > $O/cfg4.log 2>&1; grep -v amdgpu $O/cfg4.log
