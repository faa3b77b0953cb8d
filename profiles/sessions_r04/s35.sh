#!/bin/bash
# round 4, session 35: the whole GPU suite + smoke on the final binary (after s33's store-policy change)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s35; mkdir -p $O
( timeout 480 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
