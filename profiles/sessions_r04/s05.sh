#!/bin/bash
# round 4, session 5: store-segment microbenchmark (is the short segment per store instruction what holds the small
# transform sizes and the resampler's output?), native FIR gradients, the three-stage LUFS class, full-size counters
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s05; mkdir -p $O
./tools/micro/segbench 2>&1 | tee $O/segbench.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fir_family_native_gradient or lufs or loudness or resample_f16" 2>&1 | tail -6 ) > $O/pytest_sub.log 2>&1
tail -3 $O/pytest_sub.log
timeout 200 python tools/kbench.py --what lufs,lufs3 --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/kbench_lufs.log
timeout 200 python tools/rsbench.py --iters 20 --rounds 2 --only f16,f16d5 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
export PMC_FILTER="istft|tiled|stft_mel"
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh $O/pmc_istft_$c $c -- python $GRAFT_REPO_ROOT/tools/kbench.py --what istft,stft --iters 3 2>&1 | grep -v amdgpu.ids | tee -a $O/pmc_fullsize.log
  bash tools/pmc.sh $O/pmc_4096_$c $c -- python $GRAFT_REPO_ROOT/tools/kbench.py --what stft,genmel,istft --iters 3 --sr 96000 --nfft 4096 --batch 256 2>&1 | grep -v amdgpu.ids | tee -a $O/pmc_fullsize.log
done
