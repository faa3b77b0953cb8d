"""Round-3 golden fixtures (oracle/make_golden.py --round3): STFT-domain edits, apply_ir with DRR + EQ, mix, and
two loader-backed transforms, all produced by the UNMODIFIED reference on CPU.  The same checks run against the
package's CPU path here (no GPU needed) and against the HIP path on the GPU box (`-m gpu`): the GPU tests of these
rows are then ONE hop from the reference instead of HIP -> package CPU path -> reference."""
import os

import numpy as np
import pytest
import torch

import audiotools_amd as A
from audiotools_amd import transforms as tfm
from tests.test_api_parity import BankLoader

G = os.path.join(os.path.dirname(__file__), "golden")
SR = 16000
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _rows_err(got, ref):
    got, ref = got.detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if torch.is_complex(ref):
        got, ref = torch.view_as_real(got), torch.view_as_real(ref)
    g, r = got.reshape(got.shape[0] * got.shape[1], -1), ref.reshape(got.shape[0] * got.shape[1], -1)
    return float(((g - r).abs().amax(1) / r.abs().amax(1).clamp_min(1e-30)).max())


def _fresh(x, device):
    s = A.AudioSignal(torch.from_numpy(x).clone(), SR).to(device)
    s.stft(512, 128, "hann")
    return s


@pytest.mark.parametrize("device", DEVICES)
def test_spectral_edits_against_reference_goldens(device, monkeypatch):
    d = np.load(os.path.join(G, "edits_r03.npz"))
    x = d["x"]
    t = lambda k: torch.from_numpy(d[k])
    assert _rows_err(_fresh(x, device).mask_frequencies(t("fmin"), t("fmax")).stft_data, d["mask_frequencies"]) < 1e-4
    assert _rows_err(_fresh(x, device).mask_frequencies(t("fmin"), t("fmax"), val=0.25).stft_data, d["mask_frequencies_val"]) < 1e-4
    assert _rows_err(_fresh(x, device).mask_timesteps(t("tmin"), t("tmax")).stft_data, d["mask_timesteps"]) < 1e-4
    assert _rows_err(_fresh(x, device).mask_low_magnitudes(t("db_cutoff")).stft_data, d["mask_low_magnitudes"]) < 1e-4
    assert _rows_err(_fresh(x, device).shift_phase(t("shift")).stft_data, d["shift_phase"]) < 1e-4
    # corrupt_phase: feed the reference's recorded draw back (the device RNG stream differs by construction)
    noise = t("corrupt_noise")
    monkeypatch.setattr(torch, "randn_like", lambda ref, *a, **k: noise.to(ref.device))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: noise.to(k.get("device", "cpu")))
    got = _fresh(x, device).corrupt_phase(0.5).stft_data
    monkeypatch.undo()
    assert _rows_err(got, d["corrupt_phase"]) < 1e-4
    # the SpectralTransform round trip: stft -> edit -> istft
    y = _fresh(x, device).mask_frequencies(t("fmin"), t("fmax")).istft().audio_data
    assert _rows_err(y, d["masked_istft"]) < 1e-4


@pytest.mark.parametrize("device", DEVICES)
def test_apply_ir_and_mix_against_reference_goldens(device):
    d = np.load(os.path.join(G, "ir_mix_loader_r03.npz"))
    sig = lambda k: A.AudioSignal(torch.from_numpy(d[k]).clone(), SR).to(device)
    y = sig("x").apply_ir(sig("ir"), drr=torch.from_numpy(d["drr"]), ir_eq=torch.from_numpy(d["ir_eq"])).audio_data
    assert _rows_err(y, d["apply_ir"]) < 1e-4
    assert _rows_err(sig("x").apply_ir(sig("ir")).audio_data, d["apply_ir_plain"]) < 1e-4
    other = sig("other")
    y = sig("x").mix(other, snr=torch.from_numpy(d["snr"]), other_eq=torch.from_numpy(d["other_eq"])).audio_data
    assert _rows_err(y, d["mix"]) < 1e-4
    assert other.signal_length == d["x"].shape[-1]          # padded in place (effects.py:53-55)


@pytest.mark.parametrize("device", DEVICES)
def test_loader_transforms_against_reference_goldens(device):
    d = np.load(os.path.join(G, "ir_mix_loader_r03.npz"))
    states = [int(v) for v in d["states"]]
    sig = A.AudioSignal(torch.from_numpy(d["xs"]).clone(), SR)
    bn = tfm.BackgroundNoise(loader=BankLoader(A, torch.from_numpy(d["bank_noise"]), SR), snr=("uniform", 5.0, 15.0), n_bands=3)
    kw = bn.batch_instantiate(states, sig)
    y = bn(sig.clone().to(device), **A.util.prepare_batch(kw, device)).audio_data
    assert _rows_err(y, d["background_noise"]) < 1e-4
    rir = tfm.RoomImpulseResponse(loader=BankLoader(A, torch.from_numpy(d["bank_rir"]), SR), duration=0.25,
                                  drr=("uniform", 5.0, 20.0), n_bands=4)
    kw = rir.batch_instantiate(states, sig)
    y = rir(sig.clone().to(device), **A.util.prepare_batch(kw, device)).audio_data
    assert _rows_err(y, d["room_impulse_response"]) < 1e-4
