#!/bin/bash
# round 6, session 3: the shipped build with the pipelined mel stage (AT_STFT_V2_PIPE=3 is now the default): GPU suite, the
# north-star line (pool off: plain allocations are the primary figure), first RCCL contact (--force-nccl)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s03; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --force-nccl > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 6000 $O/bench.json
grep -i "rccl\|nccl" $O/bench.err | head -20
