#!/bin/bash
# round 2, GPU session 36: inverse STFT with the last pass in registers -- parity subset + timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/s36; mkdir -p $O
( timeout 200 python -m pytest tests -m gpu -x -q -k "istft or autograd or adjoint or roundtrip or loss or stretch or vocoder" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
{
timeout 100 python tools/kbench.py --what istft --iters 30
timeout 100 python tools/kbench.py --what istft --iters 50 --batch 64
} > $O/istft.log 2>&1
tail -3 $O/pytest.log; grep -v amdgpu $O/istft.log
