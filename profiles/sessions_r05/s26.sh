#!/bin/bash
# round 5, session 26: mel stage of the several-frames-per-wave kernels: unit descriptors + weights in registers (A), + the frame loop unrolled by two (B)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5s26; mkdir -p $O
for l in rsA rsB; do ( AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/libat_$l.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_oracle_golden.py -m gpu -q -k "(mel or mfcc or golden) and not autograd_native_adjoint" 2>&1 | tail -3 ) > $O/pytest_$l.log 2>&1; tail -1 $O/pytest_$l.log; done
for round in 1 2; do
for lib in libat_rs libat_rsA libat_rsB; do
  echo "### $lib round $round"
  for cfg in "512 16000" "512 44100" "256 8000" "1024 44100" "128 8000"; do set -- $cfg
    echo "# n_fft $1 @ $2"
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib.so timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
  done
done
done > $O/ab.log 2>&1
cat $O/ab.log
