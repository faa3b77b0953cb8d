#!/bin/bash
# round 5, session 27: shipped build with the 512-byte-run stores and the register-resident mel descriptors: full GPU suite + smoke
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s27; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so | cut -c1-16
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for cfg in "512 16000" "256 8000" "1024 22050" "1024 44100" "128 8000" "64 8000" "32 8000"; do set -- $cfg
  echo "# n_fft $1 @ $2"; timeout 200 python tools/kbench.py --nfft $1 --sr $2 --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
done > $O/sizes.log 2>&1; cat $O/sizes.log
