#!/bin/bash
# round 3, session 11: inverse STFT zero page / phase rotation A/B (same box, interleaved), cfg4 chain after the copy removal
cd $GRAFT_REPO_ROOT
O=gpurun_out/s50; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "istft or roundtrip or apply_ir or cfg4 or edit" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
for rep in 1 2; do
  for v in alt_old.so libaudiotools_amd.so alt_rot.so; do
    echo "### $v" >> $O/istft_ab.txt
    AT_LIB_PATH=$L/$v timeout 120 python tools/kbench.py --what istft --iters 30 2>&1 | grep -v amdgpu >> $O/istft_ab.txt
  done
done
cat $O/istft_ab.txt
timeout 300 python tools/cfgbench.py > $O/cfg.log 2>&1; grep -v amdgpu $O/cfg.log | tail -12
