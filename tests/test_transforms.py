"""The transform protocol (instantiate -> collate -> transform under a mask) on CPU tensors:
seed-for-seed equality with the UNMODIFIED reference where it can be imported, plus the
reference's own property checks (tests/data/test_transforms.py:34-85)."""
import numpy as np
import pytest
import torch

import audiotools_amd as A
from audiotools_amd import transforms as tfm
from tests import synth

NO_LOADER = ["ClippingDistortion", "Equalizer", "Quantization", "MuLawQuantization", "NoiseFloor", "VolumeChange",
             "VolumeNorm", "Silence", "LowPass", "HighPass", "RescaleAudio", "ShiftPhase", "InvertPhase",
             "CorruptPhase", "FrequencyMask", "TimeMask", "MaskLowMagnitudes", "Smoothing", "Identity", "SpectralDenoising"]


def _signal(mod, B=1, seed=5):
    x = synth.audio_batch(B, 1, 22050, seed=seed, gaps=False)
    return mod.AudioSignal(x.clone(), 44100)


@pytest.mark.parametrize("name", NO_LOADER)
def test_seeded_equality_with_reference(reference, name):
    ours = getattr(tfm, name)()
    theirs = getattr(reference.transforms if hasattr(reference, "transforms") else reference.data.transforms, name)()
    sa, sr = _signal(A), _signal(reference)
    ka = ours.instantiate(7, sa)
    kr = theirs.instantiate(7, sr)
    assert sorted(ka[ours.name]) == sorted(kr[theirs.name])
    oa = ours(sa.clone(), **ka).audio_data
    orr = theirs(sr.clone(), **kr).audio_data
    assert oa.shape == orr.shape
    assert torch.allclose(oa, orr, atol=2e-5), float((oa - orr).abs().max())


def test_compose_batch_equality_with_reference(reference):
    rt = reference.data.transforms
    mk = lambda m: m.Compose(m.LowPass(cutoff=("choice", [4000, 8000])), m.Equalizer(), m.VolumeNorm(("uniform", -30, -20)),
                             m.Choose(m.HighPass(), m.ClippingDistortion()), m.RescaleAudio(), name="chain")
    ours, theirs = mk(tfm), mk(rt)
    B = 4
    sa, sr = _signal(A, B), _signal(reference, B)
    ka = ours.batch_instantiate(list(range(B)), sa)
    kr = theirs.batch_instantiate(list(range(B)), sr)
    oa, orr = ours(sa.clone(), **ka), theirs(sr.clone(), **kr)
    assert torch.allclose(oa.audio_data, orr.audio_data, atol=2e-5)
    # batch item 0 equals the single-item application (test_transforms.py:64-76)
    k0 = ours.instantiate(0, sa[0])
    single = ours(sa[0].clone(), **k0)
    assert torch.allclose(single.audio_data, oa.audio_data[:1], atol=2e-5)
    with ours.filter("LowPass"):
        only_lp = ours(sa.clone(), **ka)
    lp = tfm.LowPass()
    assert not torch.allclose(only_lp.audio_data, oa.audio_data)


def test_mask_and_prob():
    s = _signal(A, 6)
    t = tfm.VolumeChange(db=("const", -6.0), prob=0.5)
    kw = t.batch_instantiate(list(range(6)), s)
    mask = kw["VolumeChange"]["mask"]
    assert mask.dtype == torch.bool and 0 < int(mask.sum()) < 6
    out = t(s.clone(), **kw)
    ratio = out.audio_data.abs().amax(-1)[:, 0] / s.audio_data.abs().amax(-1)[:, 0]
    assert torch.allclose(ratio[mask], torch.full_like(ratio[mask], 10 ** (-6 / 20)), atol=1e-5)
    assert torch.allclose(ratio[~mask], torch.ones_like(ratio[~mask]))
    with pytest.raises(AssertionError):
        t(s.clone(), **{"VolumeChange": {"mask": mask}})


def test_room_impulse_response_with_tensor_loader():
    g = torch.Generator().manual_seed(3)
    bank = torch.randn(5, 1, 44100, generator=g) * torch.exp(-torch.arange(44100) / 8000.0)
    t = tfm.RoomImpulseResponse(loader=tfm.TensorLoader(bank, 44100))
    s = _signal(A, 3)
    kw = t.batch_instantiate([1, 2, 3], s)
    assert kw["RoomImpulseResponse"]["ir_signal"].shape == (3, 1, 44100)
    out = t(s.clone(), **kw)
    assert torch.allclose(out.audio_data.abs().amax(-1), s.audio_data.abs().amax(-1), rtol=1e-4)
    with pytest.raises(FileNotFoundError):       # file-backed sources go through AudioLoader(sources)
        tfm.RoomImpulseResponse(sources=["/nonexistent/irs.csv"])
    bn = tfm.BackgroundNoise(loader=tfm.TensorLoader(torch.randn(4, 1, 44100, generator=g), 44100))
    kb = bn.batch_instantiate([4, 5, 6], s)
    noisy = bn(s.clone(), **kb)
    assert not torch.allclose(noisy.audio_data, s.audio_data)


def test_ir_tools_match_reference(reference):
    """decompose_ir / alter_drr / measure_drr / apply_ir vs the unmodified reference (CPU), incl.
    its quirk that the 'Hann' window over the early span is a length-1 Hann, i.e. all ones."""
    g = torch.Generator().manual_seed(1)
    ir = torch.randn(5, 1, 24000, generator=g) * torch.exp(-torch.arange(24000) / 3000.0)
    ir[2] = torch.roll(ir[2], 30, -1)
    ir[3, :, :5] *= 50
    a, r = A.AudioSignal(ir.clone(), 48000), reference.AudioSignal(ir.clone(), 48000)
    for got, ref in zip(a.decompose_ir(), r.decompose_ir()):
        assert torch.equal(got, ref)
    assert torch.allclose(a.measure_drr(), r.measure_drr())
    assert torch.allclose(a.clone().alter_drr(10.0).audio_data, r.clone().alter_drr(10.0).audio_data, atol=1e-6)
    x = 0.1 * torch.randn(5, 1, 48000, generator=g)
    mk = lambda M: M.AudioSignal(x.clone(), 48000).apply_ir(
        M.AudioSignal(ir.clone(), 48000), drr=torch.tensor([5.0, 10, 15, 20, 0]),
        ir_eq=-torch.rand(5, 6, generator=torch.Generator().manual_seed(2)))
    assert torch.allclose(mk(A).audio_data, mk(reference).audio_data, atol=1e-6)


# ------------------------------------------------------------------ batched instantiate (round 3)
def _same_tree(a, b, path=""):
    """Nested parameter dicts equal in structure, dtype, shape and value (AudioSignals by their samples)."""
    assert type(a) is type(b) or (torch.is_tensor(a) and torch.is_tensor(b)), (path, type(a), type(b))
    if isinstance(a, dict):
        assert sorted(a) == sorted(b), (path, sorted(a), sorted(b))
        for k in a:
            _same_tree(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same_tree(x, y, f"{path}[{i}]")
    elif hasattr(a, "audio_data"):
        assert a.sample_rate == b.sample_rate and a.audio_data.shape == b.audio_data.shape, path
        assert torch.equal(a.audio_data, b.audio_data), path
    else:
        assert a.dtype == b.dtype and a.shape == b.shape, (path, a.dtype, b.dtype, a.shape, b.shape)
        assert torch.equal(a, b), path


def _everything():
    g = torch.Generator().manual_seed(5)
    rir_bank = torch.randn(5, 1, 30000, generator=g) * torch.exp(-torch.arange(30000) / 4000.0)
    nz_bank = 0.1 * torch.randn(5, 2, 80000, generator=g)
    return tfm.Compose(
        tfm.VolumeChange(), tfm.LowPass(prob=0.7), tfm.Equalizer(), tfm.ClippingDistortion(prob=0.5),
        tfm.Choose(tfm.HighPass(), tfm.Quantization(), tfm.MuLawQuantization(), weights=[0.5, 0.3, 0.2]),
        tfm.RepeatUpTo(tfm.VolumeChange(("uniform", -1.0, 0.0)), max_repeat=4),
        tfm.BackgroundNoise(loader=tfm.TensorLoader(nz_bank, 44100)),
        tfm.CrossTalk(loader=tfm.TensorLoader(nz_bank, 44100), loudness_cutoff=None),
        tfm.RoomImpulseResponse(loader=tfm.TensorLoader(rir_bank, 44100), duration=0.5),
        tfm.ShiftPhase(prob=0.4), tfm.CorruptPhase(), tfm.FrequencyMask(), tfm.TimeMask(), tfm.Smoothing(),
        tfm.NoiseFloor(), tfm.Silence(), tfm.RescaleAudio(), name="everything")


def test_batch_instantiate_equals_collated_instantiate():
    """BaseTransform.batch_instantiate builds every parameter once (parameter-major); the result -- structure,
    dtypes, values, loaded excerpts -- and every RandomState afterwards equal B collated instantiate() calls."""
    x = synth.audio_batch(5, 1, 22050, seed=2, gaps=False)
    sig = A.AudioSignal(x, 44100)
    t = _everything()
    seeds = [3, 11, 12, 400, 7]
    s_batch = [np.random.RandomState(s) for s in seeds]
    s_items = [np.random.RandomState(s) for s in seeds]
    got = t.batch_instantiate(s_batch, sig)
    want = A.util.collate([t.instantiate(st, sig) for st in s_items])
    _same_tree(got, want)
    assert all(a.rand() == b.rand() for a, b in zip(s_batch, s_items))
    # plain seeds and a one-item batch
    _same_tree(t.batch_instantiate([9], sig[0]), A.util.collate([t.instantiate(9, sig[0])]))
    # one shared state: the item-major fallback keeps the reference's draw order
    sh_a, sh_b = np.random.RandomState(1), np.random.RandomState(1)
    _same_tree(t.batch_instantiate([sh_a, sh_a, sh_a], sig), A.util.collate([t.instantiate(sh_b, sig) for _ in range(3)]))
    # the batch applies like the collated one
    out_a = t(sig.clone(), **got).audio_data
    out_b = t(sig.clone(), **want).audio_data
    assert torch.equal(out_a, out_b)


# ------------------------------------------------------------------ round 4: user overrides and the parameter-major path
def _same_tree(a, b):
    assert type(a) is type(b) or (torch.is_tensor(a) and torch.is_tensor(b)), (type(a), type(b))
    if isinstance(a, dict):
        assert a.keys() == b.keys()
        for k in a:
            _same_tree(a[k], b[k])
    elif torch.is_tensor(a):
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), (a, b)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same_tree(x, y)
    else:
        assert a == b


def test_batch_instantiate_honours_user_overrides():
    """batch_instantiate == collate(instantiate) also for subclasses that override _instantiate or the public instantiate
    (the parameter-major fast paths restate the STOCK draws and must step aside), alone and as children of Compose /
    Choose; list-valued parameters become one (B, n) tensor as tt(list) collates."""
    from audiotools_amd import transforms as tfm

    class Fixed(tfm.VolumeChange):
        def _instantiate(self, state):
            return {"db": -3.0}

    class Public(tfm.VolumeChange):
        def instantiate(self, state=None, signal=None):
            out = super().instantiate(state, signal)
            out[self.name]["db"] = out[self.name]["db"] * 0 - 7.0
            return out

    class Listy(tfm.BaseTransform):
        def _instantiate(self, state):
            return {"x": [state.rand(), 2.0 * state.rand()], "n": [1, 2, 3]}

        def _transform(self, signal, x, n):
            return signal

    class NoisyEq(tfm.Equalizer):
        def _instantiate(self, state):
            return {"eq": -0.5 * state.rand(self.n_bands)}

    sig = A.AudioSignal(torch.zeros(4, 1, 100), 16000)
    seeds = [5, 6, 7, 8]
    for make in (Fixed, Public, Listy, NoisyEq,
                 lambda: tfm.Compose(Fixed(), tfm.LowPass(), Public(prob=0.5)),
                 lambda: tfm.Choose(Listy(), Fixed(), NoisyEq()),
                 lambda: tfm.Repeat(Fixed(), 2)):
        t = make()
        want = A.util.collate([t.instantiate(s, sig) for s in seeds])
        got = t.batch_instantiate(seeds, sig)
        _same_tree(got, want)
    got = Fixed().batch_instantiate(seeds, sig)["Fixed"]["db"]
    assert torch.equal(got, torch.full((4,), -3.0))
    got = Listy().batch_instantiate(seeds, sig)["Listy"]
    assert got["x"].shape == (4, 2) and got["x"].dtype == torch.float32 and got["n"].shape == (4, 3) and got["n"].dtype == torch.int64


def test_nested_batch_instantiate_does_not_reseed_states_in_use():
    """A batch_instantiate entered while another one is still drawing from its pooled RandomStates (here: from inside a
    custom _instantiate) takes fresh generator objects -- the outer draws continue their own streams."""
    from audiotools_amd import transforms as tfm

    inner = tfm.VolumeChange()
    sig = A.AudioSignal(torch.zeros(3, 1, 100), 16000)

    class Outer(tfm.BaseTransform):
        def _instantiate(self, state):
            a = state.rand()
            nested = inner.batch_instantiate([100, 101], sig)["VolumeChange"]["db"]
            return {"a": a, "b": state.rand(), "nested": nested}

        def _transform(self, signal, a, b, nested):
            return signal

    seeds = [1, 2, 3]
    t = Outer()
    want = A.util.collate([t.instantiate(s, sig) for s in seeds])
    got = t.batch_instantiate(seeds, sig)
    _same_tree(got, want)
