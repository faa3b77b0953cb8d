"""Batched, seeded audio transforms with the reference's two-phase protocol
(``audiotools/data/transforms.py:21-265``):

1. ``kwargs = tfm.instantiate(state, signal)`` draws this transform's parameters from a
   seeded ``RandomState`` (cheap, CPU, one item) -- ``batch_instantiate`` collates a list of
   them into batched tensors;
2. ``signal = tfm(signal, **kwargs)`` applies them to a (device) batch, only to the items
   whose ``mask`` is set.

Parameters live under ``kwargs[tfm.name]``; ``Compose`` prefixes child names with their
position so nested dictionaries never collide.  Every hot transform bottoms out in an
``AudioSignal`` method that runs on the HIP kernels (``low_pass`` / ``high_pass`` /
``equalizer`` -> ``at_fir_per_item_f32``; ``apply_ir`` / ``convolve`` -> ``at_fftconv_circ_f32``;
``normalize`` / ``mix`` -> ``at_lufs_f32``; the spectral family -> ``at_stft_mel_f32``).

Most concrete transforms are pure "draw parameters, call one method" recipes; they are built
from the declarative :class:`_Recipe` base instead of one hand-written class each.  File-backed
sources (``AudioLoader``) are outside the accelerated path: transforms that need other audio
(``BackgroundNoise``, ``CrossTalk``, ``RoomImpulseResponse``) take any ``loader`` callable with
the ``AudioLoader.__call__`` signature; :class:`TensorLoader` serves items from an in-memory
(device-resident, broadcastable) bank.
"""
import copy
import threading
from contextlib import contextmanager
from inspect import signature
from typing import List

import numpy as np
import torch

from . import kernels, util
from .signal import AudioSignal

tt = torch.tensor


def _flatten(d, prefix=(), keep_empty=False):
    """Tuple-keyed flat view of a nested dict.  Empty sub-dicts are DROPPED (flatten_dict's default ``keep_empty_types=()``,
    which is what the reference's prepare_batch round trip relies on) unless ``keep_empty``."""
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            if v:
                out.update(_flatten(v, prefix + (k,), keep_empty))
            elif keep_empty:
                out[prefix + (k,)] = v
        else:
            out[prefix + (k,)] = v
    return out


def _unflatten(d):
    out = {}
    for ks, v in d.items():
        cur = out
        for k in ks[:-1]:
            cur = cur.setdefault(k, {})
        cur[ks[-1]] = v
    return out


# the names the reference's module carries (from the flatten_dict package: tuple keys)
flatten, unflatten = _flatten, _unflatten


def _assemble(values: list):
    """The batched value ``util.collate`` builds from B per-item values of one parameter -- where every item first
    became a tensor with ``torch.tensor(v)`` (transforms.py:224-227) and the B tensors were stacked -- assembled ONCE:
    one numpy array, one tensor, with the dtype / shape ``torch.tensor`` gives a single item (python float -> float32,
    numpy float64 -> float64, python int / numpy int64 -> int64, bool -> bool)."""
    v0 = values[0]
    if isinstance(v0, AudioSignal):
        return AudioSignal.batch(values, pad_signals=True)
    if torch.is_tensor(v0):
        return torch.stack(values)
    if isinstance(v0, dict):
        return {k: _assemble([v[k] for v in values]) for k in v0}
    if isinstance(v0, (list, tuple)) and len(v0) and isinstance(v0[0], (AudioSignal, dict)):
        return [_assemble(list(col)) for col in zip(*values)]
    # (a list / tuple of plain values is ONE parameter: instantiate() makes it torch.tensor(list) -> (n,), collated (B, n))
    probe = torch.tensor(v0)
    try:
        arr = np.ascontiguousarray(np.asarray(values))
        out = torch.from_numpy(arr)
    except (TypeError, ValueError):
        return torch.stack([torch.tensor(v) for v in values])
    return out.to(probe.dtype).reshape((len(values),) + tuple(probe.shape))


_POOL = threading.local()


class _pooled_states:
    """``with _pooled_states(states) as rs``: one ``RandomState`` per item.  Integer seeds re-seed POOLED generator objects:
    ``RandomState(seed)`` costs ~220 us of object construction, ``pooled.seed(seed)`` 2 us for the identical stream.  The
    pool is per thread and handed out by depth: a ``batch_instantiate`` entered while another one is still drawing (a
    custom ``_instantiate`` or loader that batches something itself) takes the objects BEHIND the ones in use instead of
    re-seeding them.  The pooled objects never leave ``batch_instantiate`` (states handed in by the caller are used as
    they are)."""

    def __init__(self, states):
        self.states = states

    def __enter__(self):
        pool = getattr(_POOL, "pool", None)
        if pool is None:
            pool, _POOL.used = [], 0
            _POOL.pool = pool
        self.base = used = _POOL.used
        out = []
        for s in self.states:
            if isinstance(s, (int, np.integer)) and not isinstance(s, bool):
                if used == len(pool):
                    pool.append(np.random.RandomState(0))
                st = pool[used]
                used += 1
                st.seed(int(s))
                out.append(st)
            else:
                out.append(util.random_state(s))
        _POOL.used = used
        return out

    def __exit__(self, *exc):
        _POOL.used = self.base
        return False


def _draw(dist: tuple, state):
    """``util.sample_from_dist(dist, state)`` -- the same value from the same draws -- without the ~12 us of argument
    checking ``RandomState.choice`` spends per call: a uniform choice from a sequence is ``a[randint(0, len(a))]``
    (numpy's own legacy implementation, mtrand.pyx ``choice`` with ``p=None``)."""
    kind = dist[0]
    if kind == "const":
        return dist[1]
    if kind == "choice" and len(dist) == 2 and not np.isscalar(dist[1]):
        arr = np.asarray(dist[1])
        if arr.ndim == 1 and arr.shape[0] > 0:
            return arr[state.randint(0, arr.shape[0])]
    return getattr(state, kind)(*dist[1:])


class BaseTransform:
    """Base of all transforms.  Subclasses implement ``_instantiate(state[, signal]) -> dict``
    and ``_transform(signal, **params) -> signal``; the parameter names of ``_transform`` (plus
    ``mask``) are the keys the transform expects in its kwargs."""

    def __init__(self, keys: list = [], name: str = None, prob: float = 1.0):
        own = [k for k in signature(self._transform).parameters.keys() if k not in ("signal", "kwargs")]
        self.keys = list(keys) + own + ["mask"]
        self.prob = prob
        self.name = self.__class__.__name__ if name is None else name

    def _prepare(self, batch: dict):
        sub = batch[self.name]
        for k in self.keys:
            assert k in sub.keys(), f"{k} not in batch"
        return sub

    def _transform(self, signal):
        return signal

    def _instantiate(self, state, signal: AudioSignal = None):
        return {}

    @staticmethod
    def apply_mask(batch: dict, mask: torch.Tensor):
        return _unflatten({k: v[mask] for k, v in _flatten(batch).items()})

    @staticmethod
    def _select(batch: dict, idx: torch.Tensor, idx_host: torch.Tensor):
        """``apply_mask`` with the selected item indices already known on the host: ``v[mask]`` with a
        device mask is a nonzero() + synchronisation PER VALUE; an index gather is neither.  Host
        twins of small parameter tensors follow the selection."""
        out = {}
        for k, v in _flatten(batch).items():
            sel = v[idx]
            if torch.is_tensor(v):
                h = util.host_copy(v)
                if h is not None and h is not v:
                    util.attach_host(sel, h[idx_host])
            out[k] = sel
        return _unflatten(out)

    def transform(self, signal: AudioSignal, **kwargs):
        params = self._prepare(kwargs)
        mask = params["mask"]
        hmask = util.host_values(mask)         # no synchronisation when prepare_batch() moved the batch
        if bool(hmask.any()):
            if bool(hmask.all()) and mask.ndim == 1 and mask.numel() == signal.batch_size:
                # Every item selected: transform in place instead of the reference's gather ->
                # transform -> scatter copy of the whole batch and of every parameter
                # (transforms.py:159-164).  The cache semantics of that round trip are reproduced:
                # AudioSignal.__setitem__ only writes _loudness / stft_data back when BOTH sides have
                # them (audio_signal.py:1672-1679).
                params = {k: v for k, v in params.items() if k != "mask"}
                loud0, stft0 = signal._loudness, signal._stft_data      # raw attribute: a pending edit stays pending
                out = self._transform(signal, **params)
                loud1, stft1 = out._loudness, out._stft_data
                if out is not signal:
                    # rebind instead of copying into the old storage (a 2 GB pass at cfg4): like the
                    # methods that return ``self`` with a fresh ``audio_data`` tensor (low_pass,
                    # equalizer ...), the transformed samples live in a new tensor
                    signal.audio_data = out.audio_data
                signal._loudness = None if loud0 is None else (loud0 if loud1 is None else loud1)
                signal._stft_data = None if stft0 is None else (stft0 if stft1 is None else stft1)
                if out is not signal:
                    signal._pending_edit = out._pending_edit if signal._stft_data is stft1 else None
            elif mask.ndim == 0:
                # un-batched use (instantiate() of one item): the reference's scalar-mask indexing
                params = self.apply_mask(params, mask)
                params = {k: v for k, v in params.items() if k != "mask"}
                signal[mask] = self._transform(signal[mask], **params)
            else:
                idx_host = hmask.reshape(-1).nonzero()[:, 0]
                idx = idx_host.to(mask.device, non_blocking=True)
                params = self._select(params, idx, idx_host)
                params = {k: v for k, v in params.items() if k != "mask"}
                signal[idx] = self._transform(signal[idx], **params)
        return signal

    def __call__(self, *args, **kwargs):
        return self.transform(*args, **kwargs)

    def instantiate(self, state=None, signal: AudioSignal = None):
        state = util.random_state(state)
        needs_signal = "signal" in set(signature(self._instantiate).parameters.keys())
        params = self._instantiate(state, **({"signal": signal} if needs_signal else {}))
        for k, v in list(params.items()):
            if not isinstance(v, (AudioSignal, torch.Tensor, dict)):
                params[k] = tt(v)
        params["mask"] = tt(state.rand() <= self.prob)
        return {self.name: params}

    def batch_instantiate(self, states: list = None, signal: AudioSignal = None):
        """Parameters of B items (transforms.py:228-265).  Same result as collating B ``instantiate`` calls -- every
        item's ``RandomState`` sees the same draws in the same order -- but built parameter-major: the scalar draws
        stay per item (numpy), every parameter becomes ONE tensor (:func:`_assemble`), and transforms that own a
        loader fetch their B excerpts with one ``loader.batch`` call (no per-item tensor, no per-item launch)."""
        with _pooled_states(states) as rs:
            if len({id(st) for st in rs}) < len(rs):
                # items sharing one RandomState (or the global one): only the item-major order reproduces the draws
                return util.collate([self.instantiate(st, signal) for st in rs])
            return {self.name: self._params_for(rs, signal)}

    def _route(self) -> str:
        """Which construction describes THIS object.  A specialised ``_draw_items`` / ``_batch_params`` restates the
        ``_instantiate`` of the class that defines it, so it is stale as soon as a subclass overrides ``_instantiate``
        below it: "generic" then (BaseTransform's own parameter-major path, which calls ``self._instantiate``).  An
        overridden public ``instantiate`` can do anything: "collate" (B ``instantiate`` calls, the reference's own
        batch_instantiate, transforms.py:228-265)."""
        cls = type(self)
        if cls.instantiate is not BaseTransform.instantiate:
            return "collate"
        mro = cls.__mro__
        first = lambda name: next(i for i, c in enumerate(mro) if name in c.__dict__)
        i_inst = first("_instantiate")
        for name in ("_draw_items", "_batch_params"):
            i_fast = first(name)
            if mro[i_fast] is not BaseTransform and i_fast > i_inst:
                return "generic"
        return "fast"

    def _params_for(self, states: list, signal: AudioSignal = None):
        """This transform's parameter dict for B distinct states (also what Compose / Choose ask their children for:
        child-major order gives every item's state the same draws as the item-major loop)."""
        route = self._route()
        if route == "fast":
            return self._batch_params(states, signal)
        if route == "generic":
            items = BaseTransform._draw_items(self, states, signal)
            params = {k: _assemble([d[k] for d in items]) for k in items[0]}
            params["mask"] = _assemble([st.rand() <= self.prob for st in states])
            return params
        return util.collate([self.instantiate(st, signal) for st in states])[self.name]

    def _draw_items(self, states: list, signal: AudioSignal = None):
        """B raw parameter dicts (python / numpy values, tensors, AudioSignals), one ``_instantiate`` per state."""
        needs_signal = "signal" in set(signature(self._instantiate).parameters.keys())
        extra = {"signal": signal} if needs_signal else {}
        return [self._instantiate(st, **extra) for st in states]

    def _batch_params(self, states: list, signal: AudioSignal = None):
        items = self._draw_items(states, signal)
        params = {k: _assemble([d[k] for d in items]) for k in items[0]}
        params["mask"] = _assemble([st.rand() <= self.prob for st in states])
        return params


def _deferring(signal, body):
    """Run ``body`` with the signal's STFT-domain edits DEFERRED: between the stft() and the istft() of a spectral
    transform a per-item mask / phase shift is recorded on the signal and applied by the inverse kernel as it loads
    the spectrum (filters._spec_edit, csrc/istft.hip EDIT) instead of in a pass of its own."""
    prev = signal._defer_edits
    signal._defer_edits = True
    try:
        return body()
    finally:
        signal._defer_edits = prev


class Identity(BaseTransform):
    """Does nothing (audiotools/data/transforms.py:268-271)."""


class SpectralTransform(BaseTransform):
    """STFT before, inverse STFT after (transforms.py:274-286)."""

    def transform(self, signal, **kwargs):
        signal.stft()
        _deferring(signal, lambda: BaseTransform.transform(self, signal, **kwargs))
        signal.istft()
        return signal


class Compose(BaseTransform):
    """Apply child transforms in sequence; children are renamed ``"{position}.{name}"``
    (transforms.py:289-424)."""

    def __init__(self, *transforms: list, name: str = None, prob: float = 1.0):
        if isinstance(transforms[0], list):
            transforms = transforms[0]
        for i, t in enumerate(transforms):
            t.name = f"{i}.{t.name}"
        keys = [t.name for t in transforms]
        super().__init__(keys=keys, name=name, prob=prob)
        self.transforms = transforms
        self.transforms_to_apply = keys

    @contextmanager
    def filter(self, *names: list):
        """Temporarily restrict which children run (matched by substring of their name)."""
        old = self.transforms_to_apply
        self.transforms_to_apply = names
        yield
        self.transforms_to_apply = old

    def _transform(self, signal, **kwargs):
        for t in self.transforms:
            if any(x in t.name for x in self.transforms_to_apply):
                signal = t(signal, **kwargs)
        return signal

    def _instantiate(self, state, signal: AudioSignal = None):
        params = {}
        for t in self.transforms:
            params.update(t.instantiate(state, signal=signal))
        return params

    def _batch_params(self, states: list, signal: AudioSignal = None):
        # child-major: every item's state still sees child 0, child 1, ... in order
        params = {t.name: t._params_for(states, signal) for t in self.transforms}
        params["mask"] = _assemble([st.rand() <= self.prob for st in states])
        return params

    def __getitem__(self, idx):
        return self.transforms[idx]

    def __len__(self):
        return len(self.transforms)

    def __iter__(self):
        return iter(self.transforms)


class Choose(Compose):
    """Exactly one child (drawn per item with ``weights``) is applied (transforms.py:427-475)."""

    def __init__(self, *transforms: list, weights: list = None, name: str = None, prob: float = 1.0):
        super().__init__(*transforms, name=name, prob=prob)
        n = len(self.transforms)
        self.weights = np.array([1 / n] * n if weights is None else weights)

    def _instantiate(self, state, signal: AudioSignal = None):
        kwargs = super()._instantiate(state, signal)
        pick = state.choice(list(range(len(self.transforms))), p=self.weights)
        one_hot = []
        for i, t in enumerate(self.transforms):
            if kwargs[t.name]["mask"].item():
                kwargs[t.name]["mask"] = tt(i == pick)
            one_hot.append(kwargs[t.name]["mask"])
        kwargs["one_hot"] = one_hot
        return kwargs

    def _batch_params(self, states: list, signal: AudioSignal = None):
        params = {t.name: t._params_for(states, signal) for t in self.transforms}
        n = len(self.transforms)
        picks = np.asarray([st.choice(list(range(n)), p=self.weights) for st in states])
        one_hot = []
        for i, t in enumerate(self.transforms):
            chosen = params[t.name]["mask"] & torch.from_numpy(picks == i)      # a set mask becomes (i == pick)
            params[t.name]["mask"] = chosen
            one_hot.append(chosen)
        params["one_hot"] = torch.stack(one_hot, dim=1)      # per item torch.tensor([masks]) -> (n,), collated (B, n)
        params["mask"] = _assemble([st.rand() <= self.prob for st in states])
        return params


class Repeat(Compose):
    """The same transform ``n_repeat`` times (transforms.py:478-500)."""

    def __init__(self, transform, n_repeat: int = 1, name: str = None, prob: float = 1.0):
        super().__init__([copy.copy(transform) for _ in range(n_repeat)], name=name, prob=prob)
        self.n_repeat = n_repeat


class RepeatUpTo(Choose):
    """The transform repeated 1 .. max_repeat-1 times, count drawn per item (transforms.py:503-528)."""

    def __init__(self, transform, max_repeat: int = 5, weights: list = None, name: str = None, prob: float = 1.0):
        super().__init__([Repeat(transform, n_repeat=n) for n in range(1, max_repeat)], name=name, prob=prob,
                         weights=weights)
        self.max_repeat = max_repeat


# -------------------------------------------------------------------- recipes
class _Recipe(BaseTransform):
    """Declarative transform: ``DISTS`` maps parameter name -> default distribution tuple
    (overridable through same-named constructor kwargs), ``METHOD`` is the AudioSignal method
    called with the drawn parameters, ``FIXED`` are extra constructor kwargs forwarded to the
    method as-is."""

    DISTS = {}
    FIXED = {}
    METHOD = None
    RENAME = {}     # transform parameter name -> method keyword
    PROB = 1.0
    BASE = BaseTransform

    def __init__(self, *args, name: str = None, prob: float = None, **kwargs):
        names = list(self.DISTS) + list(self.FIXED)
        given = dict(zip(names, args))
        given.update(kwargs)
        unknown = set(given) - set(names)
        if unknown:
            raise TypeError(f"{type(self).__name__}: unexpected arguments {sorted(unknown)}")
        for k, v in {**self.DISTS, **self.FIXED}.items():
            setattr(self, k, given.get(k, v))
        BaseTransform.__init__(self, keys=list(self.DISTS), name=name, prob=self.PROB if prob is None else prob)

    def _instantiate(self, state):
        return {k: util.sample_from_dist(getattr(self, k), state) for k in self.DISTS}

    def _draw_items(self, states, signal=None):
        dists = [(k, getattr(self, k)) for k in self.DISTS]
        return [{k: _draw(d, st) for k, d in dists} for st in states]

    def _transform(self, signal, **kwargs):
        kw = {self.RENAME.get(k, k): v for k, v in kwargs.items()}
        kw.update({k: getattr(self, k) for k in self.FIXED})
        return getattr(signal, self.METHOD)(**kw)


class ClippingDistortion(_Recipe):
    """Clip to the ``perc`` quantiles (transforms.py:531-561)."""
    DISTS = {"perc": ("uniform", 0.0, 0.1)}
    METHOD = "clip_distortion"
    RENAME = {"perc": "clip_percentile"}


class Quantization(_Recipe):
    """Linear quantisation to ``channels`` levels (transforms.py:603-633)."""
    DISTS = {"channels": ("choice", [8, 32, 128, 256, 1024])}
    METHOD = "quantization"
    RENAME = {"channels": "quantization_channels"}


class MuLawQuantization(_Recipe):
    """Mu-law quantisation (transforms.py:636-666)."""
    DISTS = {"channels": ("choice", [8, 32, 128, 256, 1024])}
    METHOD = "mulaw_quantization"
    RENAME = {"channels": "quantization_channels"}


class VolumeChange(_Recipe):
    """Gain in dB (transforms.py:941-970)."""
    DISTS = {"db": ("uniform", -12.0, 0.0)}
    METHOD = "volume_change"


class VolumeNorm(_Recipe):
    """Normalise every item to ``db`` LUFS (transforms.py:973-1003)."""
    DISTS = {"db": ("const", -24)}
    METHOD = "normalize"


class LowPass(_Recipe):
    """Per-item windowed-sinc low-pass (transforms.py:1095-1131)."""
    DISTS = {"cutoff": ("choice", [4000, 8000, 16000])}
    FIXED = {"zeros": 51}
    METHOD = "low_pass"
    RENAME = {"cutoff": "cutoffs"}


class HighPass(_Recipe):
    """Per-item windowed-sinc high-pass (transforms.py:1134-1170)."""
    DISTS = {"cutoff": ("choice", [50, 100, 250, 500, 1000])}
    FIXED = {"zeros": 51}
    METHOD = "high_pass"
    RENAME = {"cutoff": "cutoffs"}


class _SpectralRecipe(_Recipe):
    """A recipe wrapped in stft() ... istft() like SpectralTransform."""

    def transform(self, signal, **kwargs):
        signal.stft()
        _deferring(signal, lambda: BaseTransform.transform(self, signal, **kwargs))
        signal.istft()
        return signal


class ShiftPhase(_SpectralRecipe):
    """Add a constant to the STFT phase (transforms.py:1200-1229)."""
    DISTS = {"shift": ("uniform", -np.pi, np.pi)}
    METHOD = "shift_phase"


class InvertPhase(ShiftPhase):
    """Phase shift of pi (transforms.py:1232-1247)."""
    DISTS = {"shift": ("const", np.pi)}


class MaskLowMagnitudes(_SpectralRecipe):
    """Zero bins below ``db_cutoff`` (transforms.py:1372-1402)."""
    DISTS = {"db_cutoff": ("uniform", -10, 10)}
    METHOD = "mask_low_magnitudes"


class RescaleAudio(BaseTransform):
    """Scale down items that peak above ``val`` (transforms.py:1173-1197)."""

    def __init__(self, val: float = 1.0, name: str = None, prob: float = 1):
        super().__init__(name=name, prob=prob)
        self.val = val

    def _transform(self, signal):
        return signal.ensure_max_of_audio(self.val)


class Silence(BaseTransform):
    """Replace the item by digital silence, keeping its cached loudness (transforms.py:1066-1092)."""

    def __init__(self, name: str = None, prob: float = 0.1):
        super().__init__(name=name, prob=prob)

    def _transform(self, signal):
        loud = signal._loudness
        signal = AudioSignal(torch.zeros_like(signal.audio_data), sample_rate=signal.sample_rate,
                             stft_params=signal.stft_params)
        signal._loudness = loud
        return signal


class Equalizer(BaseTransform):
    """Random mel-band attenuation curve ``-eq_amount * U(0,1)^n_bands`` (transforms.py:564-600)."""

    def __init__(self, eq_amount: tuple = ("const", 1.0), n_bands: int = 6, name: str = None, prob: float = 1.0):
        super().__init__(name=name, prob=prob)
        self.eq_amount = eq_amount
        self.n_bands = n_bands

    def _instantiate(self, state):
        return {"eq": -util.sample_from_dist(self.eq_amount, state) * state.rand(self.n_bands)}

    def _draw_items(self, states, signal=None):
        return [{"eq": -_draw(self.eq_amount, st) * st.rand(self.n_bands)} for st in states]

    def _transform(self, signal, eq):
        return signal.equalizer(eq)


class SpectralDenoising(Equalizer):
    """Denoise with :class:`audiotools_amd.ml.layers.SpectralGate` against a random, equalised
    white-noise clip normalised to ``nz_volume`` LUFS (transforms.py:1539-1592)."""

    def __init__(self, eq_amount: tuple = ("const", 1.0), denoise_amount: tuple = ("uniform", 0.8, 1.0),
                 nz_volume: float = -40, n_bands: int = 6, n_freq: int = 3, n_time: int = 5, name: str = None,
                 prob: float = 1):
        super().__init__(eq_amount=eq_amount, n_bands=n_bands, name=name, prob=prob)
        from .ml.layers import SpectralGate

        self.nz_volume = nz_volume
        self.denoise_amount = denoise_amount
        self.spectral_gate = SpectralGate(n_freq, n_time)

    def _instantiate(self, state):
        kwargs = super()._instantiate(state)
        kwargs["denoise_amount"] = util.sample_from_dist(self.denoise_amount, state)
        kwargs["nz"] = AudioSignal(state.randn(22050), 44100)
        return kwargs

    def _transform(self, signal, nz, eq, denoise_amount):
        nz = nz.normalize(self.nz_volume).equalizer(eq)
        self.spectral_gate = self.spectral_gate.to(signal.device)
        return self.spectral_gate(signal, nz, denoise_amount)


class NoiseFloor(BaseTransform):
    """Add white noise normalised to ``db`` LUFS (transforms.py:669-704)."""

    def __init__(self, db: tuple = ("const", -50.0), name: str = None, prob: float = 1.0):
        super().__init__(name=name, prob=prob)
        self.db = db

    def _instantiate(self, state, signal: AudioSignal):
        db = util.sample_from_dist(self.db, state)
        nz = AudioSignal(state.randn(signal.num_channels, signal.signal_length), signal.sample_rate)
        nz.normalize(db)
        return {"nz_signal": nz}

    def _transform(self, signal, nz_signal):
        return signal + nz_signal


class GlobalVolumeNorm(BaseTransform):
    """Bring the FILE-level loudness in ``signal.metadata["loudness"]`` to ``db`` (transforms.py:1006-1063)."""

    def __init__(self, db: tuple = ("const", -24), name: str = None, prob: float = 1.0):
        super().__init__(name=name, prob=prob)
        self.db = db

    def _instantiate(self, state, signal: AudioSignal):
        loud = signal.metadata.get("loudness")
        if loud is None or float(loud) == float("-inf"):
            return {"db": 0.0}
        return {"db": util.sample_from_dist(self.db, state) - float(loud)}

    def _transform(self, signal, db):
        return signal.volume_change(db)


# ------------------------------------------------------- sources of other audio
class TensorLoader:
    """In-memory replacement for ``AudioLoader`` (audiotools/data/datasets.py:71-136) on the
    accelerated path: serves excerpts from a bank of signals that already sit in memory (e.g.
    an impulse-response bank broadcast to every GPU).  ``bank`` is an AudioSignal (items along
    the batch axis) or a (N, C, T) tensor with ``sample_rate``."""

    def __init__(self, bank, sample_rate: int = None, weights: List[float] = None):
        self.bank = bank if isinstance(bank, AudioSignal) else AudioSignal(bank, sample_rate)
        self.weights = None if weights is None else np.asarray(weights, dtype=np.float64) / np.sum(weights)

    def __call__(self, state, sample_rate: int, duration: float, loudness_cutoff: float = -40,
                 num_channels: int = 1, offset: float = None, **kwargs):
        idx = int(state.choice(self.bank.batch_size, p=self.weights))
        sig = self.bank[idx: idx + 1].clone()
        n = int(duration * sig.sample_rate)
        if offset is None:
            hi = max(sig.signal_length - n, 0)
            start = int(state.randint(0, hi + 1)) if hi > 0 else 0
        else:
            start = int(offset * sig.sample_rate)
        sig.audio_data = sig.audio_data[..., start: start + n]
        if num_channels == 1:
            sig = sig.to_mono()
        sig = sig.resample(sample_rate)
        if duration is not None:
            sig = sig.zero_pad_to(int(duration * sample_rate))
        return {"signal": sig, "source_idx": 0, "item_idx": idx, "source": "tensor", "path": ""}


def _tensor_loader_batch(self, states, sample_rate: int, duration: float, loudness_cutoff: float = -40,
                         num_channels: int = 1, offset: float = None, **kwargs):
    """B items of a :class:`TensorLoader` at once: the per-item draws of ``__call__`` (numpy), then ONE gather from
    the bank, one mix-down, one resample, one pad."""
    bank = self.bank
    sr_b, T_b = bank.sample_rate, bank.signal_length
    n = int(duration * sr_b)
    hi = max(T_b - n, 0)
    idx, starts = [], []
    for st in states:
        # choice(N) without weights is randint(0, N) (numpy legacy choice); same stream, a tenth of the call cost
        idx.append(int(st.randint(0, bank.batch_size)) if self.weights is None else int(st.choice(bank.batch_size, p=self.weights)))
        if offset is None:
            starts.append(int(st.randint(0, hi + 1)) if hi > 0 else 0)
        else:
            starts.append(int(offset * sr_b))
    data = bank.audio_data
    dev = data.device
    it = torch.as_tensor(idx, device=dev)
    if len(set(starts)) == 1:
        seg = data.index_select(0, it)[..., starts[0]: starts[0] + n]
    else:                                   # random starts never run past the end (hi = T - n): a plain gather
        pos = torch.as_tensor(starts, device=dev)[:, None] + torch.arange(min(n, T_b), device=dev)[None, :]
        seg = torch.gather(data.index_select(0, it), 2, pos[:, None, :].expand(-1, data.shape[1], -1))
    sig = AudioSignal(seg.clone() if seg.data_ptr() == data.data_ptr() else seg, sr_b)
    if num_channels == 1:
        sig = sig.to_mono()
    sig = sig.resample(sample_rate)
    if duration is not None:
        sig = sig.zero_pad_to(int(duration * sample_rate))
    B = len(idx)
    return {"signal": sig, "source_idx": [0] * B, "item_idx": idx, "source": ["tensor"] * B, "path": [""] * B}


TensorLoader.batch = _tensor_loader_batch


def _loader_batch(loader, states, *args, **kwargs):
    """``loader.batch(states, ...)["signal"]`` when the loader has a batched form, else B calls collated like
    ``util.collate`` does (any callable with the AudioLoader.__call__ signature is a valid loader)."""
    if hasattr(loader, "batch"):
        return loader.batch(states, *args, **kwargs)["signal"]
    return AudioSignal.batch([loader(st, *args, **kwargs)["signal"] for st in states], pad_signals=True)


def _resolve_loader(sources, weights, loader, who):
    """``loader`` (any callable with the AudioLoader.__call__ signature, e.g. a TensorLoader over an
    HBM-resident bank) wins; otherwise the reference's construction ``AudioLoader(sources, weights)``
    (transforms.py:773, 836, 917) -- CSV files / folders through the optional soundfile package, or
    lists of ``mem://`` sources."""
    if loader is not None:
        return loader
    if sources is None:
        return None
    from .data.datasets import AudioLoader

    return AudioLoader(sources, weights)


class BackgroundNoise(BaseTransform):
    """Mix in another signal at ``snr`` dB with a random EQ (transforms.py:707-792)."""

    def __init__(self, snr: tuple = ("uniform", 10.0, 30.0), sources: List[str] = None, weights: List[float] = None,
                 eq_amount: tuple = ("const", 1.0), n_bands: int = 3, name: str = None, prob: float = 1.0,
                 loudness_cutoff: float = None, loader=None):
        super().__init__(name=name, prob=prob)
        self.snr, self.eq_amount, self.n_bands = snr, eq_amount, n_bands
        self.loader = _resolve_loader(sources, weights, loader, "BackgroundNoise")
        self.loudness_cutoff = loudness_cutoff

    def _instantiate(self, state, signal: AudioSignal):
        eq = -util.sample_from_dist(self.eq_amount, state) * state.rand(self.n_bands)
        snr = util.sample_from_dist(self.snr, state)
        bg = self.loader(state, signal.sample_rate, duration=signal.signal_duration,
                         loudness_cutoff=self.loudness_cutoff, num_channels=signal.num_channels)["signal"]
        return {"eq": eq, "bg_signal": bg, "snr": snr}

    def _batch_params(self, states, signal: AudioSignal = None):
        eq = [-_draw(self.eq_amount, st) * st.rand(self.n_bands) for st in states]
        snr = [_draw(self.snr, st) for st in states]
        # every state is its own stream: per item the order is still eq, snr, loader draws, mask
        bg = _loader_batch(self.loader, states, signal.sample_rate, duration=signal.signal_duration,
                           loudness_cutoff=self.loudness_cutoff, num_channels=signal.num_channels)
        return {"eq": _assemble(eq), "bg_signal": bg, "snr": _assemble(snr),
                "mask": _assemble([st.rand() <= self.prob for st in states])}

    def _transform(self, signal, bg_signal, snr, eq):
        return signal.mix(bg_signal.clone(), snr, eq)


class CrossTalk(BaseTransform):
    """Mix in another talker and restore the original loudness (transforms.py:795-854)."""

    def __init__(self, snr: tuple = ("uniform", 0.0, 10.0), sources: List[str] = None, weights: List[float] = None,
                 name: str = None, prob: float = 1.0, loudness_cutoff: float = -40, loader=None):
        super().__init__(name=name, prob=prob)
        self.snr = snr
        self.loader = _resolve_loader(sources, weights, loader, "CrossTalk")
        self.loudness_cutoff = loudness_cutoff

    def _instantiate(self, state, signal: AudioSignal):
        snr = util.sample_from_dist(self.snr, state)
        other = self.loader(state, signal.sample_rate, duration=signal.signal_duration,
                            loudness_cutoff=self.loudness_cutoff, num_channels=signal.num_channels)["signal"]
        return {"crosstalk_signal": other, "snr": snr}

    def _batch_params(self, states, signal: AudioSignal = None):
        snr = [_draw(self.snr, st) for st in states]
        other = _loader_batch(self.loader, states, signal.sample_rate, duration=signal.signal_duration,
                              loudness_cutoff=self.loudness_cutoff, num_channels=signal.num_channels)
        return {"crosstalk_signal": other, "snr": _assemble(snr),
                "mask": _assemble([st.rand() <= self.prob for st in states])}

    def _transform(self, signal, crosstalk_signal, snr):
        loudness = signal.loudness()
        mix = signal.mix(crosstalk_signal.clone(), snr)
        mix.normalize(loudness)
        return mix


class RoomImpulseResponse(BaseTransform):
    """Convolve with a room impulse response with random EQ and direct-to-reverberant ratio
    (transforms.py:857-938)."""

    def __init__(self, drr: tuple = ("uniform", 0.0, 30.0), sources: List[str] = None, weights: List[float] = None,
                 eq_amount: tuple = ("const", 1.0), n_bands: int = 6, name: str = None, prob: float = 1.0,
                 use_original_phase: bool = False, offset: float = 0.0, duration: float = 1.0, loader=None):
        super().__init__(name=name, prob=prob)
        self.drr, self.eq_amount, self.n_bands = drr, eq_amount, n_bands
        self.use_original_phase = use_original_phase
        self.loader = _resolve_loader(sources, weights, loader, "RoomImpulseResponse")
        self.offset, self.duration = offset, duration

    def _instantiate(self, state, signal: AudioSignal = None):
        eq = -util.sample_from_dist(self.eq_amount, state) * state.rand(self.n_bands)
        drr = util.sample_from_dist(self.drr, state)
        ir = self.loader(state, signal.sample_rate, offset=self.offset, duration=self.duration,
                         loudness_cutoff=None, num_channels=signal.num_channels)["signal"]
        ir.zero_pad_to(signal.sample_rate)
        return {"eq": eq, "ir_signal": ir, "drr": drr}

    def _batch_params(self, states, signal: AudioSignal = None):
        eq = [-_draw(self.eq_amount, st) * st.rand(self.n_bands) for st in states]
        drr = [_draw(self.drr, st) for st in states]
        ir = _loader_batch(self.loader, states, signal.sample_rate, offset=self.offset, duration=self.duration,
                           loudness_cutoff=None, num_channels=signal.num_channels)
        ir.zero_pad_to(signal.sample_rate)
        return {"eq": _assemble(eq), "ir_signal": ir, "drr": _assemble(drr),
                "mask": _assemble([st.rand() <= self.prob for st in states])}

    def _transform(self, signal, ir_signal, drr, eq):
        # The reference clones the impulse responses because apply_ir edits its argument in place
        # (transforms.py:1297).  With device-resident samples every one of those edits produces a NEW tensor
        # (per-item FIR, alter_drr kernel), so a second OBJECT over the same samples protects ``ir_signal``
        # just as well and saves the copy; the padding of the throw-away object to the signal's length is
        # skipped with it (0.45 ms of the 5.6 ms apply_ir at cfg4).
        if kernels.is_native(ir_signal.audio_data) and kernels.is_native(signal.audio_data):
            scratch = type(ir_signal)(ir_signal.audio_data, ir_signal.sample_rate, stft_params=ir_signal.stft_params)
            return signal._apply_ir(scratch, drr, eq, self.use_original_phase, True)
        return signal.apply_ir(ir_signal.clone(), drr, eq, use_original_phase=self.use_original_phase)


class Smoothing(BaseTransform):
    """Convolve with a short window, keep the input peak (transforms.py:1405-1453)."""

    def __init__(self, window_type: tuple = ("const", "average"),
                 window_length: tuple = ("choice", [8, 16, 32, 64, 128, 256, 512]), name: str = None, prob: float = 1):
        super().__init__(name=name, prob=prob)
        self.window_type, self.window_length = window_type, window_length

    def _instantiate(self, state, signal: AudioSignal = None):
        wt = util.sample_from_dist(self.window_type, state)
        wl = util.sample_from_dist(self.window_length, state)
        window = signal.get_window(window_type=wt, window_length=wl, device="cpu")
        return {"window": AudioSignal(window, signal.sample_rate)}

    def _transform(self, signal, window):
        sscale = signal.audio_data.abs().max(dim=-1, keepdim=True).values
        sscale = torch.where(sscale == 0.0, torch.ones_like(sscale), sscale)
        out = signal.convolve(window)
        oscale = out.audio_data.abs().max(dim=-1, keepdim=True).values
        oscale = torch.where(oscale == 0.0, torch.ones_like(oscale), oscale)
        return out * (sscale / oscale)


class CorruptPhase(SpectralTransform):
    """Add Gaussian noise to the STFT phase (transforms.py:1250-1278)."""

    def __init__(self, scale: tuple = ("uniform", 0, np.pi), name: str = None, prob: float = 1):
        super().__init__(name=name, prob=prob)
        self.scale = scale

    def _instantiate(self, state, signal: AudioSignal = None):
        scale = util.sample_from_dist(self.scale, state)
        corruption = state.normal(scale=scale, size=signal.phase.shape[1:])
        return {"corruption": corruption.astype("float32")}

    def _transform(self, signal, corruption):
        return signal.shift_phase(shift=corruption)


class FrequencyMask(SpectralTransform):
    """Zero a band of relative width ``f_width`` around ``f_center`` (transforms.py:1281-1324)."""

    def __init__(self, f_center: tuple = ("uniform", 0.0, 1.0), f_width: tuple = ("const", 0.1), name: str = None,
                 prob: float = 1):
        super().__init__(name=name, prob=prob)
        self.f_center, self.f_width = f_center, f_width

    def _instantiate(self, state, signal: AudioSignal):
        c = util.sample_from_dist(self.f_center, state)
        w = util.sample_from_dist(self.f_width, state)
        nyq = signal.sample_rate / 2
        return {"fmin_hz": nyq * max(c - (w / 2), 0.0), "fmax_hz": nyq * min(c + (w / 2), 1.0)}

    def _transform(self, signal, fmin_hz: float, fmax_hz: float):
        return signal.mask_frequencies(fmin_hz=fmin_hz, fmax_hz=fmax_hz)


class TimeMask(SpectralTransform):
    """Zero a time span of relative width ``t_width`` around ``t_center`` (transforms.py:1327-1369)."""

    def __init__(self, t_center: tuple = ("uniform", 0.0, 1.0), t_width: tuple = ("const", 0.025), name: str = None,
                 prob: float = 1):
        super().__init__(name=name, prob=prob)
        self.t_center, self.t_width = t_center, t_width

    def _instantiate(self, state, signal: AudioSignal):
        c = util.sample_from_dist(self.t_center, state)
        w = util.sample_from_dist(self.t_width, state)
        dur = signal.signal_duration
        return {"tmin_s": dur * max(c - (w / 2), 0.0), "tmax_s": dur * min(c + (w / 2), 1.0)}

    def _transform(self, signal, tmin_s: float, tmax_s: float):
        return signal.mask_timesteps(tmin_s=tmin_s, tmax_s=tmax_s)


def _fill_masked_with_noise(signal):
    """transforms.py:1478-1494 / 1520-1536: the bins a mask zeroed (magnitude AND phase 0) get
    magnitude and phase drawn from N(0, 1).  On the HIP path the two torch.randn_like draws are kept
    (same generator consumption as the reference) and everything else -- magnitude, phase, the hole
    mask, two indexed assignments and two polar recombinations -- is one tiled kernel."""
    from . import kernels

    X = signal.stft_data
    if kernels.spec_native(X):
        mag_r = torch.randn(X.shape, dtype=torch.float32, device=X.device)
        phase_r = torch.randn(X.shape, dtype=torch.float32, device=X.device)
        signal.stft_data = kernels.spec_polar_elem(X, phase_r, mag_r)
        return signal
    mag, phase = signal.magnitude, signal.phase
    hole = (mag == 0.0) * (phase == 0.0)
    mag = torch.where(hole, torch.randn_like(mag), mag)
    phase = torch.where(hole, torch.randn_like(phase), phase)
    signal.magnitude = mag
    signal.phase = phase
    return signal


class TimeNoise(TimeMask):
    """Like TimeMask, but the hole is filled with noise (transforms.py:1456-1495)."""

    def _transform(self, signal, tmin_s: float, tmax_s: float):
        return _fill_masked_with_noise(signal.mask_timesteps(tmin_s=tmin_s, tmax_s=tmax_s, val=0.0))


class FrequencyNoise(FrequencyMask):
    """Like FrequencyMask, but the hole is filled with noise (transforms.py:1498-1536)."""

    def _transform(self, signal, fmin_hz: float, fmax_hz: float):
        return _fill_masked_with_noise(signal.mask_frequencies(fmin_hz=fmin_hz, fmax_hz=fmax_hz))


def __getattr__(name):
    # the reference's module also carries the loader class (``from .datasets import AudioLoader``); resolved on first use
    # here because data.datasets imports this module
    if name == "AudioLoader":
        from .data.datasets import AudioLoader
        return AudioLoader
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
