#!/bin/bash
# round 2, GPU session 34: pipelined mel rounds -- parity subset, interleaved A/B, LUFS / iSTFT check
cd $GRAFT_REPO_ROOT
O=gpurun_out/s34; mkdir -p $O
( timeout 200 python -m pytest tests -m gpu -x -q -k "mel or istft or smoke or lufs or loudness" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
{
timeout 150 python tools/stftsweep.py --cfg 16:0:0,72:0:1,72:0:3,72:0:9,72:0:0,431:0:1,36:0:1,72:0:1
timeout 150 python tools/stftsweep.py --batch 64 --iters 30 --reps 5 --cfg 16:0:0,18:0:1,54:0:1,18:0:3,27:0:1
timeout 100 python tools/kbench.py --what stft,stftmel --iters 20
timeout 100 python tools/kbench.py --what stft,stftmel --iters 40 --batch 64
timeout 150 python tools/ntbench.py base
} > $O/stft.log 2>&1
tail -3 $O/pytest.log; grep -v amdgpu $O/stft.log
