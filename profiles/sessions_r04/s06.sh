#!/bin/bash
# round 4, session 6: resampler outputs through LDS (1 KB runs instead of 64-byte pieces): parity + timing
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s06; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "resample and not cfg5" 2>&1 | tail -6 ) > $O/pytest_sub.log 2>&1
tail -3 $O/pytest_sub.log
timeout 200 python tools/rsbench.py --iters 20 --rounds 3 --only f16,f16dma,mfma 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
timeout 200 python tools/rsbench.py --batch 2048 --iters 5 --rounds 2 --only f16,mfma 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_2048.log
timeout 200 python tools/rsbench.py --batch 32 --iters 50 --rounds 2 --only f16,mfma 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_32.log
timeout 200 python tools/rsbench.py --old 44100 --new 48000 --seconds 10 --iters 20 --rounds 2 --only f16,mfma 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_48k.log
