#!/bin/bash
# round 5, session 23: plain-STFT wave kernels (n_fft <= 1024, several frames per wave) storing 512-byte runs from the slab
# instead of FW segments per instruction; run length 16 vs 64 groups.  Development builds.
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5s23; mkdir -p $O
( AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/libat_rs.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stft and not autograd and not adjoint" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for round in 1 2; do
for lib in libaudiotools_amd_dev libat_rs libat_rs64 libat_run64; do
  echo "### $lib round $round"
  for cfg in "512 16000" "256 8000" "1024 22050" "128 8000"; do set -- $cfg
    echo "# n_fft $1 @ $2"
    AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/$lib.so timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stft --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | tail -2
  done
done
done > $O/ab.log 2>&1
cat $O/ab.log
