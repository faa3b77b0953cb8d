#!/bin/bash
# round 6, session 15: the loudness kernel at small batches against its wave count, finer (development switch AT_LUFS_WAVES)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
for b in 64 128 256; do
for w in 0 1280 1536 1792 2048 2304 2560 3072 3584 4096 6144; do
  echo -n "batch $b AT_LUFS_WAVES=$w  "
  AT_LUFS_WAVES=$w timeout 200 python tools/kbench.py --what lufs --iters 200 --batch $b 2>&1 | grep "^lufs"
done
done
