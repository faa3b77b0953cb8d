"""Round-5 golden fixture (oracle/make_golden.py --round5): MaskLowMagnitudes and SpectralDenoising applied by the
UNMODIFIED reference to every item.  Both compare a float32 logarithm with a dB threshold, so a kernel that evaluates the
logarithm differently in the last bit may put a bin on the other side.  The contract here: 1e-4 per row
wherever no such bin can reach, and every sample that is off must lie under a frame that holds a bin within 1e-3 dB of its
threshold (computed in float64 from the oracle's STFT) -- i.e. the disagreement is proven threshold-adjacent, not merely rare."""
import os

import numpy as np
import pytest
import torch

import audiotools_amd as A
from audiotools_amd import transforms as tfm
from oracle import restate

G = os.path.join(os.path.dirname(__file__), "golden")
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
N_FFT, HOP = 2048, 512
NEAR_DB = 1e-3


def _near_frames_lowmag(x, cutoff):
    """(B, frames) bool: frames holding a bin whose log-magnitude (audio_signal.py:1463-1494, float64) is within NEAR_DB of the
    item's cutoff.  The floor `max - top_db` is taken over the whole (sub-)batch, as the reference does."""
    X = restate.stft(x.double(), N_FFT, HOP, "hann")                          # (B, C, F, N) complex128
    logm = 10.0 * torch.log10(X.abs().pow(2).clamp_min(1e-10)) - 10.0 * np.log10(max(1e-5, 1.0))
    logm = torch.maximum(logm, logm.max() - 80.0)
    near = (logm - cutoff.double().reshape(-1, 1, 1, 1)).abs() < NEAR_DB
    return near.any(dim=(1, 2)), 0


def _near_frames_gate(x, nz, eq, sr, nz_volume):
    """The same for SpectralGate (ml/layers/spectral_gate.py:93-127): signal dB against the per-bin threshold of the
    equalised, normalised noise clip, float64; a gate bit reaches 5 frames either side through the tent smoothing."""
    # (prepare_batch moved the noise clip to the device under test in place: the yardstick is computed on the host)
    noise = A.AudioSignal(nz.audio_data.detach().cpu().clone(), nz.sample_rate).normalize(nz_volume).equalizer(torch.as_tensor(eq).cpu())
    Xn = restate.stft(noise.audio_data.double(), N_FFT, HOP, "sqrt_hann")
    ndb = 20.0 * Xn.abs().clamp_min(1e-4).log10()
    thresh = ndb.mean(-1, keepdim=True) + 3.0 * ndb.std(-1, keepdim=True)
    Xs = restate.stft(x.double(), N_FFT, HOP, "sqrt_hann")
    sdb = 20.0 * Xs.abs().clamp_min(1e-4).log10()
    # the threshold itself comes out of float32 statistics on both sides: allow its rounding on top of NEAR_DB
    near = (sdb - thresh).abs() < NEAR_DB + 2e-4
    return near.any(dim=(1, 2)), 5


def _assert_localized(y, ref, near_frames, spread, name):
    y, ref = y.detach().cpu().double(), ref.double()
    B, C, T = ref.shape
    allowed = torch.zeros(B, T, dtype=torch.bool)
    for b in range(B):
        for n in torch.nonzero(near_frames[b]).flatten().tolist():
            lo = max(0, (n - spread) * HOP - N_FFT // 2)
            hi = min(T, (n + spread) * HOP + N_FFT // 2)
            allowed[b, lo:hi] = True
    scale = ref.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    bad = ((y - ref).abs() / scale > 1e-4).any(1)                             # (B, T)
    stray = bad & ~allowed
    assert not bool(stray.any()), (name, "samples off by > 1e-4 with no threshold-adjacent bin above them:",
                                   int(stray.sum()), "of", int(bad.sum()), "bad samples")
    return int(bad.sum()), int(near_frames.sum())


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", ["MaskLowMagnitudes", "SpectralDenoising"])
def test_threshold_transforms_against_reference_golden(name, device):
    d = np.load(os.path.join(G, "transforms_r05.npz"))
    sr = int(d["sample_rate"])
    states = [int(v) for v in d["states"]]
    x = torch.from_numpy(d["x"])
    t = getattr(tfm, name)(prob=1.0)
    sig = A.AudioSignal(x.clone(), sr)
    kw = t.batch_instantiate(states, sig)
    assert bool(kw[name]["mask"].all())
    torch.manual_seed(7)
    y = t(sig.clone().to(device), **A.util.prepare_batch(kw, device)).audio_data
    ref = torch.from_numpy(d[name])
    assert float((ref - x).abs().max()) > 1e-3, "the fixture must hold transformed items"
    if name == "MaskLowMagnitudes":
        near, spread = _near_frames_lowmag(x, torch.as_tensor(kw[name]["db_cutoff"]).cpu())
    else:
        near, spread = _near_frames_gate(x, kw[name]["nz"], kw[name]["eq"], sr, t.nz_volume)
    n_bad, n_near = _assert_localized(y, ref, near, spread, name)
    print(f"{name} on {device}: {n_bad} samples beyond 1e-4, all under the {n_near} frames that hold a bin within {NEAR_DB} dB of its threshold")
    # and the disagreement stays what a handful of flipped bins can do
    far = ((y.cpu() - ref).abs() > 1e-3 * ref.abs().max()).float().mean()
    assert float(far) < 1e-3, (name, float(far))
