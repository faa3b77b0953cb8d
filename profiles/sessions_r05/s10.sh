#!/bin/bash
# round 5, session 10: checkpoint -- the whole GPU suite + smoke on the shipped library (no switches), then the three bench lines
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s10; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_ns.log 2>&1; tail -1 $O/bench_ns.log | cut -c1-1500
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1; tail -1 $O/bench_cfg4.log | cut -c1-400
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1; tail -1 $O/bench_cfg5.log | cut -c1-400
