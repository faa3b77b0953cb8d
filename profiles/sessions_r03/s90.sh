#!/bin/bash
# round 3, session 50: full GPU suite + smoke + bench line on the last binary (after the paired inverse kernel)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s90; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1500
timeout 200 python tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/k.log
