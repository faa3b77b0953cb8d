#!/bin/bash
# round 3, session 23: resampler ws kernel, all operands of a tile read before the first MFMA (one workgroup per CU, 170 registers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s62; mkdir -p $O
for rep in 1 2; do
for d in 0 1; do
  echo "### AT_RESAMPLE_DEEP=$d"
  AT_RESAMPLE_DEEP=$d timeout 120 python tools/cfgbench.py --only cfg5 2>&1 | grep "cfg5 resample"
done
done > $O/resample.log 2>&1
cat $O/resample.log
AT_RESAMPLE_DEEP=1 timeout 250 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_resample_mfma_and_valu_kernels_agree or test_resample_structured_inputs" 2>&1 | tail -3
