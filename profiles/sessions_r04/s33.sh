#!/bin/bash
# round 4, session 33: non-temporal spectrum stores in the hand-addressed forward tile of n_fft 4096 / 8192 (its input is
# fetched 1.7 x: do the write-allocated output lines push the overlapping samples out of L2?), interleaved A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s33; mkdir -p $O
for rep in 1 2; do
for v in base tnt; do
  if [ $v = base ]; then unset AT_LIB_PATH; else export AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_$v/libat.so; fi
  echo "## $v 4096 @ 96 kHz" | tee -a $O/kbench.log
  timeout 200 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr 96000 --nfft 4096 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
  echo "## $v 8192 @ 192 kHz" | tee -a $O/kbench.log
  timeout 200 python tools/kbench.py --what stft,genmel --iters 10 --batch 128 --sr 192000 --nfft 8192 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
done
done
