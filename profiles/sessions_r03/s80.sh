#!/bin/bash
# round 3, session 40: per-item gain kernel (at_scale_rows_f32) instead of torch's broadcast multiply
cd $GRAFT_REPO_ROOT
O=gpurun_out/s80; mkdir -p $O
( timeout 500 python -m pytest tests -m gpu -q -x -k "per_item_gain or arithmetic or alter_drr or apply_ir or cfg4 or room or RoomImpulse or golden or drr or transform or normalize or volume or ensure_max" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 200 python - <<'P' 2>&1 | tee $O/micro.log
import torch, time
from audiotools_amd import kernels
x = torch.randn(1024, 1, 96000, device="cuda"); g = torch.rand(1024, 1, 1, device="cuda")
out = torch.empty_like(x)
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tb = t(lambda: torch.mul(x, g, out=out)); tk = t(lambda: kernels.scale_rows(x, g, out=out)); ti = t(lambda: kernels.scale_rows(x, g, out=x))
by = 2 * x.numel() * 4
print(f"(1024,1,96000)*(1024,1,1): torch broadcast {tb:.3f} ms ({by/tb/1e9:.2f} TB/s)  at_scale_rows {tk:.3f} ms ({by/tk/1e9:.2f} TB/s)  in place {ti:.3f} ms")
P
timeout 200 python tools/cfgbench.py --only applyir,chain 2>&1 | grep "cfg4 Room\|cfg4 full\|throughput" | tee $O/cfg.log
