#!/bin/bash
# round 3, session 3: composite-radix passes (rowconv, colfft, generic STFT), tiled generic STFT + fused banded mel
cd $GRAFT_REPO_ROOT
O=gpurun_out/s42; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 400 python -m pytest tests -m gpu -q -x -k "generic or fourstep or convol or apply_ir or room or ir_tools or golden or effects or fir or sinc" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
for old in 0 1; do
  echo "### AT_STFT_GENERIC_OLD=$old  96 kHz n_fft 4096 B=256x2x10s"
  AT_STFT_GENERIC_OLD=$old timeout 120 python tools/kbench.py --what stft --iters 10 --batch 256 --sr 96000 --nfft 4096
done > $O/generic.log 2>&1
timeout 120 python tools/kbench.py --what genmel --iters 10 --batch 256 --sr 96000 --nfft 4096 >> $O/generic.log 2>&1
timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 128 --sr 192000 --nfft 8192 >> $O/generic.log 2>&1
grep -v amdgpu $O/generic.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg4prof -o k -- python $R/tools/cfgbench.py --only chain,applyir > $R/$O/cfg4.log 2>&1
f=$(find $R/$O/cfg4prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/cfg4_kernel_stats.csv
rm -rf $R/$O/cfg4prof
cd $R; grep -v amdgpu $O/cfg4.log | tail -8; head -14 $O/cfg4_kernel_stats.csv | cut -c1-150
