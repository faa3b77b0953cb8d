#!/bin/bash
# round 3, session 29: full GPU suite with the final kernels, smoke, bench lines of the three configurations, rows,
# rocprofv3 stats of the transform kernels (2048 fused, 4096 tiled forward / inverse)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s68; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1200
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1; tail -1 $O/bench_cfg4.log | cut -c1-700
timeout 300 python bench.py --config cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.log 2>&1; tail -1 $O/bench_cfg5.log | cut -c1-700
bash tools/rows.sh > $O/rows.txt 2>&1; grep -v "^E2026\|^W2026" $O/rows.txt | head -90
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/k512 -o k -- python $R/tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 --batch 512 > $R/$O/k512.log 2>&1
f=$(find $R/$O/k512 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r03_kernels_b512_kernel_stats.csv; rm -rf $R/$O/k512
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kg -o k -- python $R/tools/kbench.py --what stft,genmel,istft --iters 10 --batch 256 --sr 96000 --nfft 4096 > $R/$O/kg.log 2>&1
f=$(find $R/$O/kg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r03_generic_4096_kernel_stats.csv; rm -rf $R/$O/kg
head -6 $R/$O/r03_kernels_b512_kernel_stats.csv | cut -c1-170; head -5 $R/$O/r03_generic_4096_kernel_stats.csv | cut -c1-170
