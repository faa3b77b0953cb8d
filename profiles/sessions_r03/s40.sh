#!/bin/bash
# round 3, session 1: static-store-count STFT / iSTFT + LDS-DMA LUFS -- parity subset, A/B timings, bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/s40; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests -m gpu -q -x -k "stft or istft or loud or lufs or mel or abi or smoke or meter" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( timeout 60 python tools/lufskab.py 512 44100; timeout 60 python tools/lufskab.py 64 16000; timeout 60 python tools/lufskab.py 64 48000 ) > $O/lufs.log 2>&1
grep -v "amdgpu" $O/lufs.log
( timeout 120 python tools/stftsweep.py --batch 512 --mel 1 --iters 20 --reps 5 --cfg 72:0:1,72:0:33,72:0:0,72:0:32 ;
  timeout 120 python tools/stftsweep.py --batch 512 --mel 0 --iters 20 --reps 5 --cfg 72:0:0,72:0:32,72:0:1,72:0:33 ;
  timeout 60 python tools/stftsweep.py --batch 64 --mel 1 --iters 20 --reps 5 --cfg 72:0:1,72:0:33 ) > $O/stft.log 2>&1
grep -v "amdgpu" $O/stft.log
for lib in libaudiotools_amd.so libaudiotools_amd_r02.so libaudiotools_amd.so libaudiotools_amd_r02.so; do
  echo "### $lib"; AT_LIB_PATH=$R/audiotools_amd/lib/$lib timeout 60 python tools/kbench.py --what istft,stftmel,lufs --iters 20 --batch 512
done > $O/kbench.log 2>&1
grep -v "amdgpu" $O/kbench.log
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -2 $O/bench.log | cut -c1-1500
