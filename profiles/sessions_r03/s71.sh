#!/bin/bash
# round 3, session 32: LUFS without the per-call memset of the hop table (parity on a dirty workspace), 64-item share again
cd $GRAFT_REPO_ROOT
O=gpurun_out/s71; mkdir -p $O
( timeout 500 python -m pytest tests -m gpu -q -x -k "lufs or loud or meter or golden or salient or normalize or cfg3 or mix" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, '.')
import audiotools_amd as A
from audiotools_amd import kernels
torch.manual_seed(0)
# poison the allocator's free blocks, then measure on buffers that are recycled garbage
for T in (441000, 44100 * 3 + 7 * 4, 17640, 8000, 4410 * 5):
    x = torch.randn(6, 2, T, device='cuda') * 0.1
    ref = kernels.integrated_loudness(x, 44100).clone()
    for rep in range(3):
        junk = torch.full((64 * 1024 * 1024,), float('nan'), device='cuda'); del junk
        out = kernels.integrated_loudness(x, 44100)
        assert torch.equal(out, ref), (T, out, ref)
print('dirty-workspace LUFS: bit-stable')
PY
for rep in 1 2; do
  timeout 200 python bench.py --batch 64 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['ms_per_step'], d['kernels_ms']['stft_mel'], d['kernels_ms']['lufs_total'])"
done
