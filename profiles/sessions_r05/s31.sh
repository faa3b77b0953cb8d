#!/bin/bash
# round 5, session 31: the whole GPU suite + smoke on the final tree (binary bfece9ec, pool with twelve candidates and a lock,
# bench with the set-up call), then the table of the other transform sizes on the product path
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s31; mkdir -p $O
sha256sum audiotools_amd/lib/libaudiotools_amd.so | cut -c1-16
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for cfg in "4096 96000 256" "8192 192000 128" "2048 44100 512"; do set -- $cfg
  echo "# n_fft $1 @ $2 B$3"; timeout 200 python tools/kbench.py --nfft $1 --sr $2 --batch $3 --what stft,istft --iters 20 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
done > $O/sizes.log 2>&1; cat $O/sizes.log
