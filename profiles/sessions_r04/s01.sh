#!/bin/bash
# round 4, session 1: the fp16-split resampler's first run (parity through the raw ABI, A/B timing against the f32 MFMA
# kernel, kernel trace), the new full-size tests, the bench line with floor_ms + parity_check
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s01; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "resample or cfg5" 2>&1 | tail -25 ) > $O/pytest_resample.log 2>&1
tail -5 $O/pytest_resample.log
timeout 300 python tools/rsbench.py --iters 20 --rounds 3 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
timeout 300 python tools/rsbench.py --batch 2048 --iters 5 --rounds 2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_2048.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3000
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_r04.py -m gpu -q -k "north_star or mfcc or mel_vs_oracle or mel_golden or generic_size_short or golden" 2>&1 | tail -15 ) > $O/pytest_new.log 2>&1
tail -5 $O/pytest_new.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats -o rs -- python $GRAFT_REPO_ROOT/tools/rsbench.py --iters 20 --rounds 2 > $O/rs_stats.log 2>&1
find $O/rs_stats -name "*kernel_stats.csv" -exec head -8 {} \;
