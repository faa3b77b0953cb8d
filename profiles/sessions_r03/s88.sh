#!/bin/bash
# round 3, session 48 (final binary of the round): full GPU suite, smoke, bench lines of the three configurations, rocprofv3 stats + PMC
# of the bench command, rows, per-kernel traces (2048 fused, 4096 tiled, cfg4 chain)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s88; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1300
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1; tail -1 $O/bench_cfg4.log | cut -c1-500
timeout 300 python bench.py --config cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.log 2>&1; tail -1 $O/bench_cfg5.log | cut -c1-500
bash tools/profile_round.sh r03 > $O/profile.log 2>&1
cp gpurun_out/profile_r03/summary.json $O/r03_bench_pmc_summary.json 2>/dev/null
cp gpurun_out/profile_r03/kernel_stats.csv $O/r03_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/profile_r03/stats gpurun_out/profile_r03/pmc_*
head -4 $O/r03_bench_kernel_stats.csv | cut -c1-200
bash tools/rows.sh > $O/rows.txt 2>&1; grep -v "^E2026\|^W2026" $O/rows.txt | head -40
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/k512 -o k -- python $R/tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 --batch 512 > $R/$O/k512.log 2>&1
f=$(find $R/$O/k512 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r03_kernels_b512_kernel_stats.csv; rm -rf $R/$O/k512
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kg -o k -- python $R/tools/kbench.py --what stft,genmel,istft --iters 10 --batch 256 --sr 96000 --nfft 4096 > $R/$O/kg.log 2>&1
f=$(find $R/$O/kg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r03_generic_4096_kernel_stats.csv; rm -rf $R/$O/kg
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/c4 -o k -- python $R/bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/c4.log 2>&1
f=$(find $R/$O/c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r03_cfg4_kernel_stats.csv; rm -rf $R/$O/c4
head -12 $R/$O/r03_cfg4_kernel_stats.csv | cut -c1-150
