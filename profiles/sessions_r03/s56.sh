#!/bin/bash
# round 3, session 17: resampler ws kernel, 1 / 2 / 3 / 4 loader waves (same box, interleaved) + parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/s56; mkdir -p $O
for rep in 1 2; do
for nl in 1 2 3 4; do
  echo "### AT_RESAMPLE_LOADERS=$nl"
  AT_RESAMPLE_LOADERS=$nl timeout 120 python tools/cfgbench.py --only cfg5 2>&1 | grep "cfg5 resample 44"
done
done > $O/resample.log 2>&1
cat $O/resample.log
AT_RESAMPLE_LOADERS=2 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "resample" 2>&1 | tail -3
