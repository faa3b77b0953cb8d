import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiotools_amd as A
from audiotools_amd import kernels, util, transforms as tfm
B, T, SR = 1024, 240000, 48000
x = (0.1 * torch.randn(B, 1, T, device="cuda")).clamp_(-1, 1)
t = tfm.LowPass(cutoff=("choice", [4000, 8000, 16000]))
kw = t.batch_instantiate(list(range(B)), A.AudioSignal(x[:1], SR))
kw = util.prepare_batch(kw, "cuda")
c = kw["LowPass"]["cutoff"]
print("cutoff", c.dtype, c.shape, c.device, "host twin:", util.host_copy(c) is not None, c[:5].tolist())
def timed(fn, label):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(3): fn()
    e1.record(); h = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    print(f"{label}: host {h:.2f} ms gpu {e0.elapsed_time(e1)/3:.2f} ms")
timed(lambda: A.AudioSignal(x, SR).low_pass(c), "low_pass(tensor from batch_instantiate)")
timed(lambda: A.AudioSignal(x, SR).low_pass(c.float()), "low_pass(float tensor)")
cf = util.attach_host(c.float(), util.host_copy(c).float())
timed(lambda: A.AudioSignal(x, SR).low_pass(cf), "low_pass(float tensor + host twin)")
timed(lambda: t(A.AudioSignal(x, SR), **kw), "LowPass transform")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    A.AudioSignal(x, SR).low_pass(c); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
