#!/bin/bash
# round 2, GPU session 33: new v2 defaults, cache-policy builds of the other streaming kernels, e2e bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/s33; mkdir -p $O
L=$PWD/audiotools_amd/lib
( timeout 200 python -m pytest tests -m gpu -x -q -k "mel or istft or smoke" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
{
timeout 150 python tools/stftsweep.py --cfg 16:0:0,72:0:1,431:0:1,72:0:0,72:0:5
timeout 100 python tools/kbench.py --what stft,stftmel --iters 20
timeout 100 python tools/kbench.py --what stft,stftmel --iters 40 --batch 64
} > $O/stft.log 2>&1
{
timeout 150 python tools/ntbench.py base
AT_LIB_PATH=$L/libaudiotools_amd_ntld.so timeout 150 python tools/ntbench.py nt-loads
AT_LIB_PATH=$L/libaudiotools_amd_ntldst.so timeout 150 python tools/ntbench.py nt-ld+st
} > $O/nt.log 2>&1
timeout 200 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1
tail -3 $O/pytest.log; grep -v amdgpu $O/stft.log; grep -v amdgpu $O/nt.log; tail -1 $O/bench.log | cut -c1-1500
