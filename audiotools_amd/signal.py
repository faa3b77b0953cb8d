"""``AudioSignal``: the batched audio container of the reference
(``audiotools/core/audio_signal.py:53-1682``) with its hot-path methods
running on hand-written gfx950 kernels.

Only the container logic lives here; the DSP methods are mixed in from
``spectral.py`` (STFT / iSTFT / mel / MFCC / magnitude / phase),
``meter.py`` (LUFS), ``filters.py`` (low/high-pass, STFT masks, OLA windows),
``fx.py`` (mix, convolve, apply_ir, normalize, equalizer, ...).

Semantics kept bug-compatible with the reference where observable
(SURVEY.md 7.3 "In-place fluent API + caches"): methods mutate and return
``self``; assigning ``audio_data`` drops the cached loudness; assigning
``stft_params`` drops ``stft_data``; ``resample`` does not touch
``stft_params``.
"""
import copy
import pathlib
import typing
import warnings

import numpy as np
import torch

from . import util
from .filters import DSPMixin
from .fx import EffectMixin, ImpulseResponseMixin
from .meter import LoudnessMixin
from .spectral import STFTParams, SpectralMixin

__all__ = ["AudioSignal", "STFTParams"]


class AudioSignal(SpectralMixin, EffectMixin, LoudnessMixin, ImpulseResponseMixin, DSPMixin):
    """Batch of audio: ``audio_data`` (B, C, T) float32, ``stft_data``
    (B, C, F, N) complex64 or None, ``sample_rate``, ``stft_params``."""

    def __init__(self, audio_path_or_array, sample_rate: int = None, stft_params: STFTParams = None,
                 offset: float = 0, duration: float = None, device: str = None):
        is_path = isinstance(audio_path_or_array, (str, pathlib.Path))
        is_array = isinstance(audio_path_or_array, np.ndarray) or torch.is_tensor(audio_path_or_array)
        if not (is_path or is_array):
            raise ValueError("audio_path_or_array must be either a Path, string, numpy array, or torch Tensor!")

        self.path_to_file = None
        self.audio_data = None
        self.sources = None
        self.stft_data = None
        if is_path:
            self.load_from_file(audio_path_or_array, offset=offset, duration=duration, device=device)
        else:
            assert sample_rate is not None, "Must set sample rate!"
            self.load_from_array(audio_path_or_array, sample_rate, device=device)
        self.window = None
        self.stft_params = stft_params
        self.metadata = {"offset": offset, "duration": duration}

    # ------------------------------------------------------------ factories
    @property
    def path_to_input_file(self):
        return self.path_to_file

    @classmethod
    def zeros(cls, duration: float, sample_rate: int, num_channels: int = 1, batch_size: int = 1, **kwargs):
        return cls(torch.zeros(batch_size, num_channels, int(duration * sample_rate)), sample_rate, **kwargs)

    @classmethod
    def excerpt(cls, audio_path, offset: float = None, duration: float = None, state=None, **kwargs):
        """Random excerpt of ``duration`` seconds between ``offset`` and the end of the source
        (audio_signal.py:180-226): ONE ``state.uniform`` draw."""
        total_duration = util.info(audio_path).duration
        state = util.random_state(state)
        lower_bound = 0 if offset is None else offset
        upper_bound = max(total_duration - duration, 0)
        offset = state.uniform(lower_bound, upper_bound)
        signal = cls(audio_path, offset=offset, duration=duration, **kwargs)
        signal.metadata["offset"] = offset
        signal.metadata["duration"] = duration
        return signal

    @classmethod
    def salient_excerpt(cls, audio_path, loudness_cutoff: float = None, num_tries: int = 8, state=None, **kwargs):
        """Random excerpt whose loudness exceeds ``loudness_cutoff`` LUFS, giving up after
        ``num_tries`` draws (audio_signal.py:228-286).

        The reference measures one candidate at a time (a full ``loudness()`` per try).  For sources
        resident on a GPU the candidates of ALL tries are measured by one batched launch: the
        offsets are drawn from a COPY of ``state``, the first candidate above the cutoff wins, and
        the real ``state`` is then advanced by exactly the draws the sequential loop would have
        made -- same excerpt, same RNG state afterwards, one launch + one synchronisation."""
        state = util.random_state(state)
        if loudness_cutoff is None:
            return cls.excerpt(audio_path, state=state, **kwargs)
        mem = util.memory_audio(audio_path)
        if mem is not None and mem[0].is_cuda and num_tries is not None and num_tries > 1:
            return cls._salient_excerpt_batched(audio_path, mem, loudness_cutoff, int(num_tries), state, **kwargs)
        loudness = -np.inf
        num_try = 0
        while loudness <= loudness_cutoff:
            excerpt = cls.excerpt(audio_path, state=state, **kwargs)
            loudness = excerpt.loudness()
            num_try += 1
            if num_tries is not None and num_try >= num_tries:
                break
        return excerpt

    @classmethod
    def _salient_excerpt_batched(cls, audio_path, mem, loudness_cutoff, num_tries, state, offset: float = None,
                                 duration: float = None, **kwargs):
        bank, sr = mem
        total_duration = bank.shape[-1] / sr
        lower_bound = 0 if offset is None else offset
        upper_bound = max(total_duration - duration, 0)
        probe = np.random.RandomState()
        probe.set_state(state.get_state())
        offsets = [probe.uniform(lower_bound, upper_bound) for _ in range(num_tries)]
        n = int(duration * sr)
        rows = []
        for o in offsets:
            seg = bank[:, int(o * sr): int(o * sr) + n]
            if seg.shape[-1] == 0:
                raise RuntimeError(f"Audio file {audio_path} with offset {o} and duration {duration} is empty!")
            if seg.shape[-1] < n:         # the last excerpt of a short source: measured as is in the reference
                rows = None
                break
            rows.append(seg)
        if rows is None:                  # ragged candidates: keep the sequential form
            loudness, num_try = -np.inf, 0
            while loudness <= loudness_cutoff:
                excerpt = cls.excerpt(audio_path, offset=offset, duration=duration, state=state, **kwargs)
                loudness = excerpt.loudness()
                num_try += 1
                if num_try >= num_tries:
                    break
            return excerpt
        cand = cls(torch.stack(rows), sr)
        loud = cand.loudness().cpu()                                   # one launch, one synchronisation
        above = (loud > loudness_cutoff).nonzero()
        pick = int(above[0]) if above.numel() else num_tries - 1
        for _ in range(pick + 1):                                      # the draws the sequential loop makes
            state.uniform(lower_bound, upper_bound)
        signal = cls(audio_path, offset=offsets[pick], duration=duration, **kwargs)
        signal.metadata["offset"] = offsets[pick]
        signal.metadata["duration"] = duration
        signal._loudness = loud[pick: pick + 1].to(signal.device)
        return signal

    @classmethod
    def wave(cls, frequency: float, duration: float, sample_rate: int, num_channels: int = 1,
             shape: str = "sine", **kwargs):
        """Simple test tone ("sine", "square", "sawtooth", "triangle")."""
        from scipy import signal as sps

        t = torch.linspace(0, duration, int(duration * sample_rate))
        ph = 2 * np.pi * frequency * t
        if shape == "sawtooth":
            wave = torch.from_numpy(sps.sawtooth(ph.numpy(), 0.5))
        elif shape == "square":
            wave = torch.from_numpy(sps.square(ph.numpy()))
        elif shape == "sine":
            wave = torch.sin(ph)
        elif shape == "triangle":
            wave = torch.from_numpy(sps.sawtooth(ph.numpy(), 0.5))
        else:
            raise ValueError(f"Invalid shape {shape}")
        wave = wave.unsqueeze(0).unsqueeze(0).repeat(1, num_channels, 1)
        return cls(wave, sample_rate, **kwargs)

    @classmethod
    def batch(cls, audio_signals: list, pad_signals: bool = False, truncate_signals: bool = False,
              resample: bool = False, dim: int = 0):
        """Concatenate signals along ``dim`` after equalising rate/length (audio_signal.py:380-470;
        RuntimeError if they differ and no policy is given).  Signals that need resampling are
        grouped by (rate, length, channels, device) and every group is ONE resampler launch;
        padding writes each signal once into the batch tensor (no per-signal F.pad + cat)."""
        rates = [s.sample_rate for s in audio_signals]
        # As in the reference (audio_signal.py:437-438) the lengths are taken BEFORE resampling: the
        # batch is as long as the longest ORIGINAL signal, and a resampled signal that ends up longer
        # is cut by its negative pad length.
        lengths = [s.signal_length for s in audio_signals]
        if len(set(rates)) != 1:
            if not resample:
                raise RuntimeError(
                    f"Not all signals had the same sample rate! Got {rates}. "
                    f"All signals must have the same sample rate, or resample must be True. ")
            groups = {}
            for s in audio_signals:
                if s.sample_rate != rates[0]:
                    key = (s.sample_rate, tuple(s.audio_data.shape[1:]), s.audio_data.device, s.audio_data.dtype)
                    groups.setdefault(key, []).append(s)
            for (rate, _, _, _), members in groups.items():
                if len(members) == 1:
                    members[0].resample(rates[0])
                    continue
                sizes = [m.batch_size for m in members]
                stacked = cls(torch.cat([m.audio_data for m in members], 0), rate).resample(rates[0])
                for m, part in zip(members, torch.split(stacked.audio_data, sizes, 0)):
                    m.audio_data = part
                    m.sample_rate = rates[0]
        if len(set(lengths)) != 1:
            if pad_signals:
                longest = max(lengths)
                if dim == 0 and len({(s.num_channels, s.audio_data.device, s.audio_data.dtype) for s in audio_signals}) == 1:
                    first = audio_signals[0].audio_data
                    total = sum(s.batch_size for s in audio_signals)
                    out_data = torch.zeros((total, first.shape[1], longest), dtype=first.dtype, device=first.device)
                    at = 0
                    for s in audio_signals:
                        n = min(s.signal_length, longest)
                        out_data[at: at + s.batch_size, :, :n] = s.audio_data[..., :n]
                        at += s.batch_size
                    # the reference pads every input signal in place (audio_signal.py:449-451); keep that observable
                    # effect with the inputs' OWN storage: one copy of the padded block, split among them (sharing
                    # the batch tensor made `batch *= g` edit the inputs)
                    own = out_data.clone()
                    at = 0
                    for s in audio_signals:
                        s.audio_data = own[at: at + s.batch_size]
                        at += s.batch_size
                    out = cls(out_data, sample_rate=audio_signals[0].sample_rate)
                    out.path_to_file = [s.path_to_file for s in audio_signals]
                    return out
                for s in audio_signals:
                    s.zero_pad(0, longest - s.signal_length)
            elif truncate_signals:
                shortest = min(lengths)
                for s in audio_signals:
                    s.truncate_samples(shortest)
            else:
                raise RuntimeError(
                    f"Not all signals had the same length! Got {lengths}. "
                    f"All signals must be the same length, or pad_signals/truncate_signals must be True. ")
        out = cls(torch.cat([s.audio_data for s in audio_signals], dim=dim), sample_rate=audio_signals[0].sample_rate)
        out.path_to_file = [s.path_to_file for s in audio_signals]
        return out

    # ------------------------------------------------------------------ I/O
    def load_from_file(self, audio_path, offset: float, duration: float = None, device: str = None):
        """Decode ``duration`` seconds from ``offset`` (audio_signal.py:473-531).  ``mem://`` sources
        registered with ``util.register_memory_audio`` are sliced where they live (host or HBM);
        real files need the optional ``soundfile`` package -- file decoding is outside the
        accelerated path."""
        mem = util.memory_audio(audio_path)
        if mem is not None:
            bank, sr = mem
            start = int(offset * sr)
            stop = bank.shape[-1] if duration is None else start + int(duration * sr)
            data = bank[:, start:stop]
            if data.shape[-1] == 0:
                raise RuntimeError(f"Audio file {audio_path} with offset {offset} and duration {duration} is empty!")
            # a COPY, as decoding a file returns fresh samples: in-place edits of the excerpt (`sig *= g`, the
            # masked write-back of BaseTransform.transform) must never reach the registered bank, and the copy
            # is 16-byte aligned whatever the start offset (the read-only candidate scan of
            # _salient_excerpt_batched keeps slicing the bank in place)
            self.audio_data = data.clone().unsqueeze(0)
        else:
            try:
                import soundfile
            except ImportError as e:
                raise RuntimeError("loading audio files needs the optional `soundfile` package") from e
            info = soundfile.info(str(audio_path))
            sr = info.samplerate
            start = int(offset * sr)
            frames = -1 if duration is None else int(duration * sr)
            data, sr = soundfile.read(str(audio_path), start=start, frames=frames, always_2d=True, dtype="float32")
            if data.shape[0] == 0:
                raise RuntimeError(f"Audio file {audio_path} with offset {offset} and duration {duration} is empty!")
            self.audio_data = torch.from_numpy(data.T.copy()).unsqueeze(0)
        self.original_signal_length = self.signal_length
        self.sample_rate = sr
        self.path_to_file = audio_path
        return self if device is None else self.to(device)

    def load_from_array(self, audio_array, sample_rate: int, device: str = "cpu"):
        data = util.ensure_tensor(audio_array)
        if data.dtype == torch.double:
            data = data.float()
        while data.ndim < 3:
            data = data.unsqueeze(0)
        self.audio_data = data
        self.original_signal_length = self.signal_length
        self.sample_rate = sample_rate
        return self.to(device)

    def write(self, audio_path):
        import soundfile

        if self.audio_data[0].abs().max() > 1:
            warnings.warn("Audio amplitude > 1 clipped when saving")
        soundfile.write(str(audio_path), self.audio_data[0].cpu().numpy().T, self.sample_rate)
        self.path_to_file = audio_path
        return self

    # -------------------------------------------------------------- copying
    def deepcopy(self):
        return copy.deepcopy(self)

    def copy(self):
        return copy.copy(self)

    def clone(self):
        twin = type(self)(self.audio_data.clone(), self.sample_rate, stft_params=self.stft_params)
        if self.stft_data is not None:
            twin.stft_data = self.stft_data.clone()
        if self._loudness is not None:
            twin._loudness = self._loudness.clone()
        twin.path_to_file = copy.deepcopy(self.path_to_file)
        twin.metadata = copy.deepcopy(self.metadata)
        return twin

    def detach(self):
        if self._loudness is not None:
            self._loudness = self._loudness.detach()
        if self.stft_data is not None:
            self.stft_data = self.stft_data.detach()
        self.audio_data = self.audio_data.detach()
        return self

    # ------------------------------------------------------ signal-level ops
    def to_mono(self):
        self.audio_data = self.audio_data.mean(1, keepdim=True)
        return self

    def to(self, device: str):
        if self._loudness is not None:
            self._loudness = self._loudness.to(device)
        if self.stft_data is not None:
            self.stft_data = self.stft_data.to(device)
        if self.audio_data is not None:
            self.audio_data = self.audio_data.to(device)
        return self

    def float(self):
        self.audio_data = self.audio_data.float()
        return self

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def numpy(self):
        return self.audio_data.detach().cpu().numpy()

    def zero_pad(self, before: int, after: int):
        self.audio_data = torch.nn.functional.pad(self.audio_data, (before, after))
        return self

    def zero_pad_to(self, length: int, mode: str = "after"):
        short = max(length - self.signal_length, 0)
        if mode == "before":
            self.zero_pad(short, 0)
        elif mode == "after":
            self.zero_pad(0, short)
        return self

    def trim(self, before: int, after: int):
        end = None if after == 0 else -after
        self.audio_data = self.audio_data[..., before:end]
        return self

    def truncate_samples(self, length_in_samples: int):
        self.audio_data = self.audio_data[..., :length_in_samples]
        return self

    # ----------------------------------------------------------- properties
    @property
    def device(self):
        if self.audio_data is not None:
            return self.audio_data.device
        if self.stft_data is not None:
            return self.stft_data.device
        return None

    @property
    def audio_data(self):
        return self._audio_data

    @audio_data.setter
    def audio_data(self, data):
        if data is not None:
            assert torch.is_tensor(data), "audio_data should be torch.Tensor"
            assert data.ndim == 3, "audio_data should be 3-dim (B, C, T)"
        self._audio_data = data
        self._loudness = None  # stale once the samples change
        self._peak_of = None   # (samples, max |.|, position) left by alter_drr for the convolution that follows

    samples = audio_data

    # A STFT-domain edit recorded by mask_frequencies / mask_timesteps / mask_low_magnitudes / shift_phase while a
    # SpectralTransform runs (``_defer_edits``): ``istft()`` applies it inside the inverse kernel, reading
    # ``stft_data`` materialises it -- either way the caller sees the reference's values (a NEW stft_data tensor
    # whose untouched bins are bit-identical), the edit's own read + write pass only happens when somebody looks.
    _pending_edit = None
    _defer_edits = False

    @property
    def stft_data(self):
        if self._pending_edit is not None:
            edit, self._pending_edit = self._pending_edit, None
            self._stft_data = edit.apply(self._stft_data)
        return self._stft_data

    @stft_data.setter
    def stft_data(self, data):
        if data is not None:
            assert torch.is_tensor(data) and torch.is_complex(data)
            if self._stft_data is not None and self._stft_data.shape != data.shape:
                warnings.warn("stft_data changed shape")
        self._pending_edit = None
        self._stft_data = data

    @property
    def batch_size(self):
        return self.audio_data.shape[0]

    @property
    def signal_length(self):
        return self.audio_data.shape[-1]

    length = signal_length

    @property
    def shape(self):
        return self.audio_data.shape

    @property
    def signal_duration(self):
        return self.signal_length / self.sample_rate

    duration = signal_duration

    @property
    def num_channels(self):
        return self.audio_data.shape[1]

    # ----------------------------------------------------------- arithmetic
    def _twin(self, op, other):
        """``self.clone()`` followed by the in-place ``op`` (audio_signal.py:1385-1415), in ONE pass over the
        samples when the result has the samples' own shape and dtype: the clone's copy of ``audio_data``
        (a read and a write of the whole batch, 0.4 ms of apply_ir's final rescale at cfg4) is the output
        buffer of the arithmetic instead of its input."""
        v = util._get_value(other)
        a = self.audio_data
        vs = tuple(v.shape) if torch.is_tensor(v) else ()
        one_pass = (torch.is_tensor(a) and not a.requires_grad and not (torch.is_tensor(v) and v.requires_grad)
                    and not torch.is_complex(a) and not (torch.is_tensor(v) and torch.is_complex(v))
                    and (not torch.is_tensor(v) or v.device == a.device or v.ndim == 0)
                    and torch.broadcast_shapes(tuple(a.shape), vs) == tuple(a.shape)
                    and not isinstance(v, complex))
        if not one_pass:
            out = self.clone()
            out.audio_data = getattr(out.audio_data, "__i%s__" % op)(v)
            return out
        res = getattr(torch, op)(a, v, out=torch.empty_like(a))
        twin = type(self)(res, self.sample_rate, stft_params=self.stft_params)
        if self.stft_data is not None:
            twin.stft_data = self.stft_data.clone()
        twin.path_to_file = copy.deepcopy(self.path_to_file)      # (no _loudness: assigning the samples resets it)
        twin.metadata = copy.deepcopy(self.metadata)
        return twin

    def __add__(self, other):
        return self._twin("add", other)

    def __iadd__(self, other):
        self.audio_data += util._get_value(other)
        return self

    def __radd__(self, other):
        return self + other

    def __sub__(self, other):
        return self._twin("sub", other)

    def __isub__(self, other):
        self.audio_data -= util._get_value(other)
        return self

    def __mul__(self, other):
        return self._twin("mul", other)

    def __imul__(self, other):
        self.audio_data *= util._get_value(other)
        return self

    def __rmul__(self, other):
        return self * other

    # ------------------------------------------------------- representation
    def _info(self):
        dur = f"{self.signal_duration:0.3f}" if self.signal_duration else "[unknown]"
        return {
            "duration": f"{dur} seconds",
            "batch_size": self.batch_size,
            "path": self.path_to_file if self.path_to_file else "path unknown",
            "sample_rate": self.sample_rate,
            "num_channels": self.num_channels if self.num_channels else "[unknown]",
            "audio_data.shape": self.audio_data.shape,
            "stft_params": self.stft_params,
            "device": self.device,
        }

    def __str__(self):
        return "\n".join(f"{k}: {v}" for k, v in self._info().items())

    __repr__ = __str__

    def markdown(self):
        """The signal's description as a two-column markdown table (audio_signal.py:1568-1597)."""
        rows = ["| Key | Value", "|---|---"] + [f"| {k} | {v} |" for k, v in self._info().items()]
        return "\n".join(rows) + "\n"

    def __rich__(self):
        """``rich.print(signal)``: the same description as a rich table (audio_signal.py:1607-1618)."""
        from rich.table import Table

        table = Table(title=type(self).__name__)
        table.add_column("Key", style="green")
        table.add_column("Value", style="cyan")
        for k, v in self._info().items():
            table.add_row(k, str(v))
        return table

    def hash(self):
        """A name for the audio CONTENT (audio_signal.py:673-703 hashes the bytes of a temporary wav file; here the
        float32 samples, their shape and the rate go through sha256 directly -- equal audio, equal hash; no file)."""
        import hashlib

        h = hashlib.sha256()
        a = self.audio_data.detach().to("cpu", torch.float32).contiguous()
        h.update(repr((tuple(a.shape), int(self.sample_rate))).encode())
        h.update(a.numpy().tobytes())
        return h.hexdigest()

    # ---------------------------------------------------- equality/indexing
    def __eq__(self, other):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                if not torch.allclose(v, other.__dict__[k], atol=1e-6):
                    print(f"Max abs error for {k}: {(v - other.__dict__[k]).abs().max()}")
                    return False
        return True

    def __ne__(self, other):
        return not self == other

    __hash__ = object.__hash__

    @staticmethod
    def _is_whole_batch_key(key):
        return torch.is_tensor(key) and key.ndim == 0 and key.item() is True

    @staticmethod
    def _is_batch_key(key):
        return isinstance(key, (bool, int, list, slice, tuple)) or (torch.is_tensor(key) and key.ndim <= 1)

    def __getitem__(self, key):
        if self._is_whole_batch_key(key):
            assert self.batch_size == 1
            audio, loud, stft = self.audio_data, self._loudness, self.stft_data
        elif self._is_batch_key(key):
            audio = self.audio_data[key]
            loud = self._loudness[key] if self._loudness is not None else None
            stft = self.stft_data[key] if self.stft_data is not None else None
        else:
            raise TypeError(f"unsupported index {key!r}")
        view = type(self)(audio, self.sample_rate, stft_params=self.stft_params)
        view._loudness = loud
        view._stft_data = stft
        view.sources = None
        return view

    def __setitem__(self, key, value):
        if not isinstance(value, type(self)):
            self.audio_data[key] = value
            return
        if self._is_whole_batch_key(key):
            assert self.batch_size == 1
            self.audio_data = value.audio_data
            self._loudness = value._loudness
            self.stft_data = value.stft_data
            return
        if self._is_batch_key(key):
            if self.audio_data is not None and value.audio_data is not None:
                self.audio_data[key] = value.audio_data
            if self._loudness is not None and value._loudness is not None:
                self._loudness[key] = value._loudness
            if self.stft_data is not None and value.stft_data is not None:
                self.stft_data[key] = value.stft_data
