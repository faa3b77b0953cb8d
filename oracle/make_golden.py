"""Generate tests/golden/*.npz from the UNMODIFIED reference (shim-imported from
/root/reference).  TEST INFRASTRUCTURE ONLY.  Run in the build container:

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden            # the round-1 fixtures
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden --round3   # edits_r03.npz, ir_mix_loader_r03.npz only
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden --round4   # transforms_r04.npz only
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden --round5   # transforms_r05.npz only

Inputs are stored next to the outputs (float16-exact values are not assumed), so the
fixtures do not depend on torch's RNG staying stable.  Every array is produced by calling
the reference's own public methods on CPU torch.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import import_reference  # noqa: E402
from tests import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    at = import_reference()
    AudioSignal, STFTParams = at.AudioSignal, at.STFTParams
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # cfg1-shaped STFT (B reduced to 2 to keep the fixture small): 1 s mono @16 kHz, 512/128 hann
    x = synth.audio_batch(2, 1, 16000, seed=101, gaps=False)
    s = AudioSignal(x.clone(), 16000)
    X = s.stft(512, 128, "hann")
    y = s.clone().istft(512, 128, "hann").audio_data
    Xm = AudioSignal(x.clone(), 16000).stft(512, 128, "sqrt_hann", match_stride=True)
    np.savez_compressed(os.path.join(OUT, "stft_cfg1.npz"), x=x.numpy(), stft=X.numpy(), istft=y.numpy(),
                        stft_match_stride_sqrt_hann=Xm.numpy())

    # cfg2-shaped STFT + mel (B=1, 2 ch, 0.5 s @44.1 kHz, default params 2048/512 hann, 80 mels)
    x = synth.audio_batch(1, 2, 22050, seed=102, gaps=False)
    s = AudioSignal(x.clone(), 44100)
    mel = s.mel_spectrogram(80)
    mfcc = AudioSignal(x.clone(), 44100).mfcc()
    np.savez_compressed(os.path.join(OUT, "mel_cfg2.npz"), x=x.numpy(), stft=s.stft_data.numpy(),
                        mel=mel.numpy(), mfcc=mfcc.numpy(),
                        mel_basis=AudioSignal.get_mel_filters(44100, 2048, 80, 0.0, None))

    # loudness: the reference's seeded batch (tests/core/test_loudness.py:31-52) + level/gap batch
    np.random.seed(0)
    arr = np.random.randn(16, 2, 16000).astype(np.float32)
    l16 = AudioSignal(torch.from_numpy(arr.copy()), 16000).loudness()
    xg = synth.audio_batch(4, 2, 4 * 16000, seed=103, sample_rate=16000)
    xg[1] = 0.0                                   # digital silence -> -70 clamp
    xg[2, :, : 16000 * 3] *= 1e-4                 # mostly below the absolute gate
    # store the float16-rounded input and the loudness of exactly that input
    xg16 = torch.from_numpy(xg.numpy().astype(np.float16).astype(np.float32))
    lg = AudioSignal(xg16.clone(), 16000).loudness()
    sine = synth.sine(1000, 44100, 20.0, amp=0.99924)
    ls = AudioSignal(sine.clone(), 44100).loudness()
    fl = {fc: AudioSignal(xg16[:2, :, : 2 * 16000].clone(), 16000).loudness(filter_class=fc).numpy()
          for fc in ("Fenton/Lee 1", "Dash et al.")}
    np.savez_compressed(os.path.join(OUT, "loudness.npz"), seeded_randn_16k=l16.numpy(),
                        gaps_x=xg.numpy().astype(np.float16), gaps_lufs=lg.numpy(), sine_1000_lufs=ls.numpy(),
                        fenton_lee_1=fl["Fenton/Lee 1"], dash=fl["Dash et al."])

    # filters / resample / equalizer / convolve on a short stereo clip @48 kHz (cfg4 family)
    x = synth.audio_batch(3, 1, 24000, seed=104, gaps=False, sample_rate=48000)
    cut = torch.tensor([4000.0, 8000.0, 16000.0])
    lp = AudioSignal(x.clone(), 48000).low_pass(cut).audio_data
    hp = AudioSignal(x.clone(), 48000).high_pass(torch.tensor([500.0, 1000.0, 2000.0])).audio_data
    db = -torch.rand(3, 6, generator=torch.Generator().manual_seed(9))
    eq = AudioSignal(x.clone(), 48000).equalizer(db).audio_data
    g = torch.Generator().manual_seed(10)
    ir = torch.randn(3, 1, 9600, generator=g) * torch.exp(-torch.arange(9600) / (0.05 * 48000))
    cv = AudioSignal(x.clone(), 48000).convolve(AudioSignal(ir.clone(), 48000)).audio_data
    rs = AudioSignal(x.clone(), 48000).resample(16000).audio_data
    xr = synth.audio_batch(2, 2, 22050, seed=105, gaps=False)
    rs2 = AudioSignal(xr.clone(), 44100).resample(16000).audio_data
    np.savez_compressed(os.path.join(OUT, "effects_cfg4.npz"), x=x.numpy(), lp_cut=cut.numpy(), low_pass=lp.numpy(),
                        high_pass=hp.numpy(), eq_db=db.numpy(), equalizer=eq.numpy(), ir=ir.numpy(),
                        convolve=cv.numpy(), resample_48k_16k=rs.numpy(), xr=xr.numpy(),
                        resample_441_16k=rs2.numpy())
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


def round3():
    """Fixtures added in round 3 ("two-hop parity"): STFT-domain edits, apply_ir with DRR + EQ, mix, and
    two loader-backed transforms, all produced by the unmodified reference.  Writes only the new files."""
    at = import_reference()
    AudioSignal = at.AudioSignal
    os.makedirs(OUT, exist_ok=True)
    from tests.test_api_parity import BankLoader

    # ---- STFT-domain edits (dsp.py:217-370) on a 0.25 s mono pair @16 kHz, 512/128 hann
    sr = 16000
    x = synth.audio_batch(2, 1, 4000, seed=301, gaps=False, sample_rate=sr)

    def fresh():
        s = AudioSignal(x.clone(), sr)
        s.stft(512, 128, "hann")
        return s

    fmin, fmax = torch.tensor([500.0, 1000.0]), torch.tensor([1500.0, 3000.0])
    tmin, tmax = torch.tensor([0.05, 0.10]), torch.tensor([0.10, 0.20])
    dbc = torch.tensor([-20.0, -30.0])
    shift = torch.tensor([0.7, -1.1])
    out = {"x": x.numpy(), "fmin": fmin.numpy(), "fmax": fmax.numpy(), "tmin": tmin.numpy(), "tmax": tmax.numpy(),
           "db_cutoff": dbc.numpy(), "shift": shift.numpy()}
    out["mask_frequencies"] = fresh().mask_frequencies(fmin, fmax).stft_data.numpy()
    out["mask_frequencies_val"] = fresh().mask_frequencies(fmin, fmax, val=0.25).stft_data.numpy()
    out["mask_timesteps"] = fresh().mask_timesteps(tmin, tmax).stft_data.numpy()
    out["mask_low_magnitudes"] = fresh().mask_low_magnitudes(dbc).stft_data.numpy()
    out["shift_phase"] = fresh().shift_phase(shift).stft_data.numpy()
    # corrupt_phase draws torch.randn_like(phase) (dsp.py:369): record the draw, the test feeds it back
    s = fresh()
    noise = {}
    real_randn_like = torch.randn_like

    def rec(t, *a, **k):
        noise["n"] = real_randn_like(t, *a, **k)
        return noise["n"]

    torch.randn_like = rec
    try:
        torch.manual_seed(5)
        out["corrupt_phase"] = s.corrupt_phase(0.5).stft_data.numpy()
    finally:
        torch.randn_like = real_randn_like
    out["corrupt_noise"] = noise["n"].numpy()
    out["masked_istft"] = fresh().mask_frequencies(fmin, fmax).istft().audio_data.numpy()   # the SpectralTransform round trip
    np.savez_compressed(os.path.join(OUT, "edits_r03.npz"), **out)

    # ---- apply_ir with DRR alteration + IR EQ (effects.py:125-179), mix with SNR + EQ (effects.py:27-64)
    x = synth.audio_batch(2, 1, 12000, seed=302, gaps=False, sample_rate=sr)
    g = torch.Generator().manual_seed(303)
    ir = torch.randn(2, 1, 4000, generator=g) * torch.exp(-torch.arange(4000) / 500.0)
    ir[:, :, 40] += 4.0                                    # a clear direct path, not at sample 0
    drr = torch.tensor([5.0, 15.0])
    ir_eq = -torch.rand(2, 6, generator=g)
    y_ir = AudioSignal(x.clone(), sr).apply_ir(AudioSignal(ir.clone(), sr), drr=drr, ir_eq=ir_eq).audio_data
    y_ir_plain = AudioSignal(x.clone(), sr).apply_ir(AudioSignal(ir.clone(), sr)).audio_data
    other = 0.05 * torch.randn(2, 1, 9000, generator=g)   # shorter than the signal: padded in place
    snr = torch.tensor([5.0, 15.0])
    other_eq = -torch.rand(2, 3, generator=g)
    y_mix = AudioSignal(x.clone(), sr).mix(AudioSignal(other.clone(), sr), snr=snr, other_eq=other_eq).audio_data
    res = {"x": x.numpy(), "ir": ir.numpy(), "drr": drr.numpy(), "ir_eq": ir_eq.numpy(), "apply_ir": y_ir.numpy(),
           "apply_ir_plain": y_ir_plain.numpy(), "other": other.numpy(), "snr": snr.numpy(), "other_eq": other_eq.numpy(),
           "mix": y_mix.numpy()}

    # ---- two loader-backed transforms (transforms.py:707-794, 857-938) over an in-memory bank, states 10..12
    T = at.data.transforms
    bank_n = (0.1 * torch.randn(4, 2, 24000, generator=g))
    bank_r = torch.randn(4, 1, 8000, generator=g) * torch.exp(-torch.arange(8000) / 900.0)
    xs = synth.audio_batch(3, 1, 16000, seed=304, gaps=False, sample_rate=sr)
    states = [10, 11, 12]
    bn = T.BackgroundNoise(sources=[], snr=("uniform", 5.0, 15.0), n_bands=3)
    bn.loader = BankLoader(at, bank_n, sr)
    sig = AudioSignal(xs.clone(), sr)
    y_bn = bn(sig.clone(), **bn.batch_instantiate(states, sig)).audio_data
    rir = T.RoomImpulseResponse(sources=[], duration=0.25, drr=("uniform", 5.0, 20.0), n_bands=4)
    rir.loader = BankLoader(at, bank_r, sr)
    y_rir = rir(sig.clone(), **rir.batch_instantiate(states, sig)).audio_data
    res.update(bank_noise=bank_n.numpy().astype(np.float32), bank_rir=bank_r.numpy(), xs=xs.numpy(),
               states=np.asarray(states), background_noise=y_bn.numpy(), room_impulse_response=y_rir.numpy())
    np.savez_compressed(os.path.join(OUT, "ir_mix_loader_r03.npz"), **res)
    for f in ("edits_r03.npz", "ir_mix_loader_r03.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


R4_TRANSFORMS = ["ClippingDistortion", "Equalizer", "Quantization", "MuLawQuantization", "NoiseFloor", "VolumeChange", "VolumeNorm",
                 "Silence", "LowPass", "HighPass", "RescaleAudio", "ShiftPhase", "InvertPhase", "FrequencyMask", "TimeMask",
                 "MaskLowMagnitudes", "Smoothing", "Identity", "SpectralDenoising"]
R4_STATES = [3, 4, 5, 6]


def round4():
    """Fixtures added in round 4 (the transform GPU tests compared the HIP path with the package's
    own CPU path, two hops from the reference): every loader-free transform of data/transforms.py, instantiated by the
    UNMODIFIED reference for states 3..6 with prob = 0.5 and applied by it on CPU.  tests/test_golden_r04.py replays the
    same states through the package (CPU path here, HIP path under -m gpu) and compares with these outputs."""
    at = import_reference()
    AudioSignal = at.AudioSignal
    T = at.data.transforms
    sr = 44100
    x = synth.audio_batch(len(R4_STATES), 1, 9000, seed=401, gaps=False, sample_rate=sr)
    out = {"x": x.numpy(), "states": np.asarray(R4_STATES), "sample_rate": np.asarray(sr)}
    for name in R4_TRANSFORMS:
        t = getattr(T, name)(prob=0.5) if name != "Identity" else T.Identity()
        sig = AudioSignal(x.clone(), sr)
        kw = t.batch_instantiate(R4_STATES, sig)
        torch.manual_seed(7)         # (transforms that draw from torch's global generator inside transform())
        y = t(sig.clone(), **kw)
        out[name] = y.audio_data.numpy()
    np.savez_compressed(os.path.join(OUT, "transforms_r04.npz"), **out)
    print("transforms_r04.npz", os.path.getsize(os.path.join(OUT, "transforms_r04.npz")) // 1024, "KiB")


R5_TRANSFORMS = ["MaskLowMagnitudes", "SpectralDenoising"]


def round5():
    """Fixture added in round 5: the two transforms that compare a float32 logarithm with a dB
    threshold, applied by the UNMODIFIED reference to EVERY item (prob = 1; with round 4's prob = 0.5 and states 3..6 the
    SpectralDenoising row happened to transform no item at all).  tests/test_golden_r05.py replays the states through the
    package and demands 1e-4 everywhere except in samples reached by a bin that lies within 1e-3 dB of its threshold."""
    at = import_reference()
    AudioSignal = at.AudioSignal
    T = at.data.transforms
    sr = 44100
    x = synth.audio_batch(len(R4_STATES), 1, 9000, seed=501, gaps=False, sample_rate=sr)
    out = {"x": x.numpy(), "states": np.asarray(R4_STATES), "sample_rate": np.asarray(sr)}
    for name in R5_TRANSFORMS:
        t = getattr(T, name)(prob=1.0)
        sig = AudioSignal(x.clone(), sr)
        kw = t.batch_instantiate(R4_STATES, sig)
        assert bool(kw[name]["mask"].all())
        torch.manual_seed(7)
        y = t(sig.clone(), **kw)
        out[name] = y.audio_data.numpy()
    np.savez_compressed(os.path.join(OUT, "transforms_r05.npz"), **out)
    print("transforms_r05.npz", os.path.getsize(os.path.join(OUT, "transforms_r05.npz")) // 1024, "KiB")


if __name__ == "__main__":
    if "--round5" in sys.argv:
        round5()
    elif "--round4" in sys.argv:
        round4()
    elif "--round3" in sys.argv:
        round3()
    else:
        main()
