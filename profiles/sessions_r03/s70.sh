#!/bin/bash
# round 3, session 31: the 64-item share (one rank of an 8-way shard): eager vs hipGraph replay, per-kernel durations
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s70; mkdir -p $O
for rep in 1 2; do
  timeout 200 python bench.py --batch 64 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['ms_per_step'], d['kernels_ms']['stft_mel'], d['kernels_ms']['lufs_total'])"
  timeout 200 python bench.py --batch 64 --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'], d['config']['launch'][:40])"
done 2>&1 | tee $O/share.log
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o k -- python $R/bench.py --batch 64 --steps 100 --warmup 10 --no-cpu-baseline --no-share > $O/p.log 2>&1
f=$(find $O/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/share64_kernel_stats.csv; rm -rf $O/p
head -6 $O/share64_kernel_stats.csv | cut -c1-200
