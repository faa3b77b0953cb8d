#!/bin/bash
# round 3, session 43: max-ilp scheduler strategy per source (longconv / firfft / fir) vs the default library
cd $GRAFT_REPO_ROOT
O=gpurun_out/s83; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
f() { grep -v -e amdgpu.ids -e "^$"; }
for rep in 1 2; do
for v in "" _ilp_longconv _ilp_firfft _ilp_fir; do
  lib=libaudiotools_amd$v.so
  echo "### rep $rep lib=$lib"
  export AT_LIB_PATH=$L/$lib
  case "$v" in
   ""|_ilp_longconv) timeout 200 python tools/convbench.py 2>&1 | f | tail -6;;
  esac
  case "$v" in
   ""|_ilp_firfft) timeout 200 python tools/firbench.py 677 2>&1 | f | tail -3; timeout 200 python tools/firbench.py 153 2>&1 | f | tail -3;;
  esac
  case "$v" in
   ""|_ilp_fir) timeout 200 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-share 2>&1 | f | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg5 ms_per_step', d['ms_per_step'], 'kernels', d.get('kernels_ms'))";;
  esac
  timeout 200 python tools/cfgbench.py --only applyir,chain 2>&1 | grep "cfg4 Room\|cfg4 full"
done; done 2>&1 | tee $O/ab.log
