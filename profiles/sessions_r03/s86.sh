#!/bin/bash
# round 3, session 46: -mllvm -amdgpu-sched-strategy=max-memory-clause on every source vs the shipped library
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s86; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
f() { grep -v -e amdgpu.ids -e "^$"; }
for rep in 1 2; do
for lib in libaudiotools_amd.so libaudiotools_amd_mmc.so; do
  echo "### rep $rep lib=$lib"
  export AT_LIB_PATH=$L/$lib
  timeout 200 python tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 2>&1 | f
  timeout 200 python tools/kbench.py --what stft,genmel,istft --iters 10 --batch 256 --sr 96000 --nfft 4096 2>&1 | f
  timeout 200 python tools/convbench.py 2>&1 | f | head -1
  timeout 200 python tools/firbench.py 677 2>&1 | f | tail -1
  timeout 200 python tools/cfgbench.py --only applyir,chain 2>&1 | grep "cfg4 Room\|cfg4 full"
  timeout 200 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-share 2>&1 | f | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg5 ms_per_step', d['ms_per_step'], 'kernels', d.get('kernels_ms'))"
done; done 2>&1 | tee $O/ab.log
