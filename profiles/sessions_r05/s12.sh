#!/bin/bash
# round 5, session 12: evidence on the final binary -- rocprofv3 --kernel-trace --stats and one --pmc pass per counter group for
# the three bench configurations (tools/profile_round.sh writes lib_sha256 into every summary), untraced bench lines in between
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s12; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so > $O/lib_sha256.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_ns.log 2>&1
timeout 900 bash tools/profile_round.sh r05_bench > $O/profile_bench.log 2>&1
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1
timeout 900 bash tools/profile_round.sh r05_cfg4 --config cfg4 > $O/profile_cfg4.log 2>&1
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1
timeout 1200 bash tools/profile_round.sh r05_cfg5 --config cfg5 > $O/profile_cfg5.log 2>&1
for t in bench cfg4 cfg5; do cp gpurun_out/profile_r05_$t/summary.json $O/r05_${t}_pmc_summary.json; cp gpurun_out/profile_r05_$t/kernel_stats.csv $O/r05_${t}_kernel_stats.csv; done
rm -rf gpurun_out/profile_r05_bench/pmc_* gpurun_out/profile_r05_cfg4/pmc_* gpurun_out/profile_r05_cfg5/pmc_* gpurun_out/profile_r05_*/stats
ls -la $O
