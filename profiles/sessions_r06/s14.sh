#!/bin/bash
# round 6, session 14: the loudness kernel at B = 512 against its wave count (development switch AT_LUFS_WAVES: segments per row)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
for w in 0 2048 3072 4096 6144 8192 12288 16384; do
  echo "### AT_LUFS_WAVES=$w"
  AT_LUFS_WAVES=$w timeout 200 python tools/kbench.py --what lufs --iters 50 2>&1 | grep "^lufs"
done
for w in 0 1024 2048 4096 8192; do
  echo "### batch 64 AT_LUFS_WAVES=$w"
  AT_LUFS_WAVES=$w timeout 200 python tools/kbench.py --what lufs --iters 100 --batch 64 2>&1 | grep "^lufs"
done
