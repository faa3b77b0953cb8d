#!/bin/bash
# round 6, session 16: the loudness kernel's new segmentation cost model (shipped build): batches 16 ... 512, the loudness tests
cd $GRAFT_REPO_ROOT
for b in 16 32 64 128 256 512; do echo -n "batch $b  "; timeout 200 python tools/kbench.py --what lufs --iters 200 --batch $b 2>&1 | grep "^lufs"; done
echo -n "batch 64 Fenton/Lee (three stages)  "; timeout 200 python tools/kbench.py --what lufs3 --iters 100 --batch 64 2>&1 | grep "^lufs"
timeout 900 python -m pytest tests -x -q -m gpu -k "loudness or lufs or north_star or cfg3 or normalize or salient or mix" 2>&1 | grep -v "^Extension modules" | tail -3
