#!/bin/bash
# round 6, session 12: HALF EXCHANGE -- the ascending butterflies of the v2 kernel's pass 3 take their inputs through
# v_permlane16/32_swap (4 x 4 transposes between the wave's rows and registers) instead of the slab: semantics test, parity, A/B
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
(cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o /tmp/plt permlane_test.hip 2>/dev/null && /tmp/plt)
echo "### parity, hx"
AT_LIB_PATH=$L/libat_hx.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stft or mel or north_star or cfg2 or cfg5" 2>&1 | grep -v "^Extension modules" | tail -3
for round in 1 2 3; do
for lib in libaudiotools_amd_dev libat_hx; do
  echo "### $lib round $round"
  AT_LIB_PATH=$L/$lib.so timeout 200 python tools/kbench.py --what stft,stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
done
done
