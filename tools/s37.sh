#!/bin/bash
# round 2, closing session: full GPU suite + per-kernel rocprofv3 rows with the final binary
cd $GRAFT_REPO_ROOT
O=gpurun_out/s37; mkdir -p $O
( timeout 320 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/k512 -o k -- python $R/tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 --batch 512 > $R/$O/k512.log 2>&1
f=$(find $R/$O/k512 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/k512_kernel_stats.csv
cd $R; tail -3 $O/pytest.log; head -6 $O/k512_kernel_stats.csv | cut -c1-160; grep -v "amdgpu\|^E2026\|^W2026\|^I2026" $O/k512.log | head -5
