#!/bin/bash
# round 3, session 33: resampler with three tile buffers, two alternating loader waves, operands in registers (A/B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s72; mkdir -p $O
AT_RESAMPLE_WS3=1 timeout 250 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_resample_mfma_and_valu_kernels_agree or test_resample_structured_inputs" 2>&1 | tail -3
for rep in 1 2 3; do
for d in 0 1; do
  echo "### AT_RESAMPLE_WS3=$d"
  AT_RESAMPLE_WS3=$d timeout 120 python tools/cfgbench.py --only cfg5 2>&1 | grep "cfg5 resample 44"
done
done > $O/resample.log 2>&1
cat $O/resample.log
