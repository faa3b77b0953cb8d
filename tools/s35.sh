#!/bin/bash
# round 2, final evidence session: full GPU tests, rocprofv3 stats + PMC of the bench command, per-row
# timings and the three bench lines -- all with the final binary
cd $GRAFT_REPO_ROOT
O=gpurun_out/s35; mkdir -p $O
( timeout 340 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
timeout 330 bash tools/profile_round.sh r02c < /dev/null > $O/round.log 2>&1
run() { echo "### $*"; timeout 120 "$@" 2>&1 < /dev/null | grep -v -e amdgpu.ids -e "^$"; }
{
run python tools/kbench.py --what stft,stftmel,lufs,istft,copy --iters 20
run python tools/kbench.py --what stft,stftmel,lufs,istft --iters 50 --batch 64
run python tools/cfgbench.py
run python tools/convbench.py
run python tools/ntbench.py final
} > $O/rows.txt 2>&1
{
timeout 200 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1
timeout 120 python bench.py --steps 10 --warmup 3 --config cfg4 2>/dev/null | tail -1
timeout 120 python bench.py --steps 10 --warmup 3 --config cfg5 --batch 256 2>/dev/null | tail -1
} > $O/bench_lines.jsonl 2>&1
tail -3 $O/pytest.log; tail -30 $O/rows.txt | cut -c1-160; cut -c1-400 $O/bench_lines.jsonl
