#!/bin/bash
# round 4, session 15: s14 again after the workspace fix (power-of-two sizes with a hop the fused kernels do not take now
# also run the one-pass path): parity of everything that inverts
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s15; mkdir -p $O
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -k "generic or istft or stretch or pitch or vocoder or spectral or edit" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -25 | cut -c1-400
for cfg in "44100 512 100" "16000 512 128" "16000 128 8"; do
  set -- $cfg
  echo "### sr=$1 n_fft=$2" | tee -a $O/kbench.log
done
