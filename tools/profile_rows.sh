#!/bin/bash
# rocprofv3 kernel stats of the per-kernel and per-config benches (GPU box).  usage: tools/profile_rows.sh <tag>
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rows_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k512 -o k -- python $R/tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 --batch 512 > $O/k512.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg -o c -- python $R/tools/cfgbench.py > $O/cfg.log 2>&1
for d in k512 cfg; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv
done
head -12 $O/k512_kernel_stats.csv | cut -c1-200
head -25 $O/cfg_kernel_stats.csv | cut -c1-200
