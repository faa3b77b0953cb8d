#!/bin/bash
# round 4, session 19: collect_windows / overlap_and_add kernels
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s19; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -k "windows or feeding or loader" > $O/pytest.log 2>&1
grep -v "^  File\|^Extension" $O/pytest.log | tail -8 | cut -c1-300
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/windows_bench.log
import torch, torch.nn.functional as F
import audiotools_amd as A
sr=44100
x=(0.1*torch.randn(64,2,10*sr,device="cuda"))
def t(fn,n=10):
    fn(); torch.cuda.synchronize()
    e=[torch.cuda.Event(enable_timing=True) for _ in range(2)]; e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize(); return e[0].elapsed_time(e[1])/n
def native():
    s=A.AudioSignal(x,sr); s.collect_windows(0.5,0.125); return s.overlap_and_add(0.125)
def torch_form():
    hop=int(0.125*sr); win=int(0.5*sr)//hop*hop
    xp=F.pad(x,(hop,hop)); Tp=xp.shape[-1]
    u=F.unfold(xp.reshape(-1,1,1,Tp),kernel_size=(1,win),stride=(1,hop))
    w=u.permute(0,2,1).reshape(-1,1,win)
    u2=w.reshape(128,-1,win).permute(0,2,1)
    kw=dict(output_size=(1,Tp),kernel_size=(1,win),stride=(1,hop))
    return (F.fold(u2,**kw)/F.fold(torch.ones_like(u2),**kw))[...,hop:-hop]
print(f"collect_windows + overlap_and_add, 64 x 2ch x 10 s, 0.5 s windows / 0.125 s hop: kernels {t(native):.3f} ms, torch unfold / fold formulation {t(torch_form):.3f} ms")
PY
