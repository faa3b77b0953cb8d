#!/bin/bash
# round 5, session 15: the whole GPU suite + smoke on the FINAL binary (after the alter_drr change)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s15; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so > $O/lib_sha256.txt
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
