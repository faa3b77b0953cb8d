#!/bin/bash
# round 3, session 37: the failing LUFS workspace test in detail; fused peak restoration of the room transform (parity, cfg4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s76; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lufs_does_not_depend" 2>&1 | grep -v "^$" | tail -40 > $O/lufs.log; tail -30 $O/lufs.log
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_rescale or cfg4 or apply_ir or room or RoomImpulse or fourstep or convol" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
for r in 0 1 0 1; do
  echo "### AT_LONGCONV_RESCALE=$r"
  AT_LONGCONV_RESCALE=$r timeout 200 python tools/cfgbench.py --only applyir,chain 2>&1 | grep "cfg4 Room\|cfg4 full"
done 2>&1 | tee $O/cfg.log
