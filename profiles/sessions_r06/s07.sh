#!/bin/bash
# round 6, session 7: the multi-rank control flow of bench.py on one device (two ranks, gloo), the backward-first process test,
# the selection that aborted in s05
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s07; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_multi_rank or backward_pass_as_the_first" 2>&1 | grep -v "^Extension modules" | tail -25
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel or stft or north_star or cfg2 or cfg5" 2>&1 | grep -v "^Extension modules" | tail -4
