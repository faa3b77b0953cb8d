#!/bin/bash
# round 4, session 2: the register-prefetch form of the fp16-split resampler (parity incl. bit-equality with the LDS-DMA
# form, three-way timing), its counters, the cfg5 bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s02; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "resample and not cfg5" 2>&1 | tail -25 ) > $O/pytest_resample.log 2>&1
tail -4 $O/pytest_resample.log
timeout 300 python tools/rsbench.py --iters 20 --rounds 3 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
timeout 300 python tools/rsbench.py --batch 2048 --iters 5 --rounds 2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_2048.log
timeout 300 python tools/rsbench.py --batch 32 --iters 50 --rounds 2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench_32.log
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1; tail -1 $O/bench_cfg5.log | cut -c1-2500
KIND=f16 bash tools/pmc_rs.sh gpurun_out/s02/pmc_f16 2>&1 | grep -v amdgpu.ids | tee $O/pmc_f16.log
