#!/bin/bash
# round 3, session 8: spectral gate kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/s47; mkdir -p $O
( timeout 300 python -m pytest tests -m gpu -q -x -k "spectral_gate or SpectralDenoising" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
tail -10 $O/pytest.log
timeout 120 python - > $O/gate.log 2>&1 <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
import audiotools_amd as A
from audiotools_amd import kernels
from audiotools_amd.ml.layers import SpectralGate
B = 256
x = (0.1 * torch.randn(B, 2, 441000, device="cuda")).clamp_(-1, 1)
nz = A.AudioSignal(0.02 * torch.randn(1, 2, 44100, device="cuda"), 44100)
sig = A.AudioSignal(x, 44100)
g = SpectralGate().to("cuda")
def timed(fn, label, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{label:40s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms", flush=True)
timed(lambda: g(sig, nz, 0.9), "SpectralGate native gate kernel")
saved = kernels.spec_native
kernels.spec_native = lambda X: False
timed(lambda: g(sig, nz, 0.9), "SpectralGate torch formulation")
kernels.spec_native = saved
X = sig.clone().stft(2048, 512, "sqrt_hann")
thr = torch.zeros(1, 2, 1025, device="cuda")
timed(lambda: kernels.spec_gate(X, thr, torch.tensor([0.9]), g.tent_f, g.tent_t), "at_spec_gate_f32 alone (3.6 GB spectrum)")
PY
grep -v amdgpu $O/gate.log
