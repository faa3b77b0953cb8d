#!/bin/bash
# round 6, session 2: what the instruction classes of the v2 kernel cost (measurement builds, results wrong by construction):
# pipe3 = baseline; abl1 no v_sqrt; abl2 no unit rounds; abl4 no magnitude writes; abl8 no thread-0 selects; abl16 no partner permute
cd $GRAFT_REPO_ROOT
export AT_DEV_KNOBS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6s02; mkdir -p $O
L=$GRAFT_REPO_ROOT/audiotools_amd/lib
{
for round in 1 2; do
for lib in libat_pipe3 libat_abl1 libat_abl2 libat_abl4 libat_abl8 libat_abl16; do
  echo "### $lib round $round"
  AT_LIB_PATH=$L/$lib.so timeout 200 python tools/kbench.py --what stftmel --iters 30 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "^pool"
done
done
echo "### pool tests (shipped library)"
unset AT_DEV_KNOBS
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "placement or caller_buffers" 2>&1 | tail -5
} > $O/ab.log 2>&1
cat $O/ab.log
