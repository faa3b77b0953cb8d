#!/bin/bash
# round 3, session 30: inverse tile at 8192 with composed third-pass twiddles (two workgroups per CU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s69; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "istft_tiled or generic_sizes_vs" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
timeout 120 python tools/kbench.py --what istft --iters 10 --batch 128 --sr 192000 --nfft 8192 2>&1 | grep istft
timeout 120 python tools/kbench.py --what istft --iters 10 --batch 256 --sr 96000 --nfft 4096 2>&1 | grep istft
done > $O/istft.log 2>&1
cat $O/istft.log
