#!/bin/bash
# round 4, session 3 (after the tile-maximum fix of the register-prefetch resampler: __builtin_bit_cast on a vector-element
# lvalue read element 0 four times): full GPU suite, three-way resampler timing, cfg5 counters
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s03; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 300 python tools/rsbench.py --iters 20 --rounds 3 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
timeout 300 python tools/rsbench.py --batch 2048 --iters 5 --rounds 2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_2048.log
bash tools/profile_round.sh r04_cfg5 --config cfg5 > $O/profile_cfg5.log 2>&1
tail -c 3000 $O/profile_cfg5.log
