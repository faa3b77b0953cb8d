#!/bin/bash
# round 6, session 1: the v2 kernel with the mel unit rounds of frame f inside frame f + 1 (AT_STFT_V2_PIPE=3, development builds, A/B)
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6s01; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
{
echo "### parity, pipe3"
AT_LIB_PATH=$L/libat_pipe3.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel_vs_oracle or mel_golden or cfg2_full or north_star or default_params_kernel or caller_buffers or mel_options or mel_loss" 2>&1 | tail -5
for round in 1 2 3; do
for lib in libaudiotools_amd_dev libat_pipe3; do
  echo "### $lib round $round"
  AT_LIB_PATH=$L/$lib.so timeout 200 python tools/kbench.py --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids
done
done
} > $O/ab.log 2>&1
cat $O/ab.log
