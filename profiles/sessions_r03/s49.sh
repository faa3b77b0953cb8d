#!/bin/bash
# round 3, session 10: rocprofv3 stats + PMC of the bench command (no share block), bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/s49; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1800
bash tools/profile_round.sh r03 > $O/profile.log 2>&1
cp gpurun_out/profile_r03/summary.json $O/r03_bench_pmc_summary.json 2>/dev/null
cp gpurun_out/profile_r03/kernel_stats.csv $O/r03_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/profile_r03/stats gpurun_out/profile_r03/pmc_*
head -4 $O/r03_bench_kernel_stats.csv | cut -c1-200
timeout 100 python tools/kbench.py --what stft,stftmel,lufs,istft,copy --iters 20 > $O/kbench.log 2>&1; grep -v amdgpu $O/kbench.log
