"""Small host-side helpers shared by the signal object and the transforms
(counterparts of the helpers the hot path uses from the reference's
``audiotools/core/util.py``: ``ensure_tensor`` :56-89, ``_get_value`` :92-96,
``random_state`` :99-126, ``seed`` :129-151, ``sample_from_dist`` :383-423,
``collate`` :426-479, ``prepare_batch`` :346-380)."""
import csv
import dataclasses
import glob
import numbers
import os
import random
import typing
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import torch


HOST_SHADOW_MAX = 1 << 16   # elements; transform parameters (masks, cutoffs, dB values ...) are tiny


def attach_host(dev_t: torch.Tensor, host_t: torch.Tensor) -> torch.Tensor:
    """Remember the CPU original of a small parameter tensor that was moved to a device.  Host-side
    decisions about it (is the mask all true? which filter length does the largest cutoff need?)
    then cost no device-to-host synchronisation: in the reference every such question is a
    ``.item()`` / boolean-mask sync in the middle of the launch stream (SURVEY.md 8(e) "scaling risk")."""
    if dev_t is not host_t and dev_t.is_cuda and not host_t.is_cuda and host_t.numel() <= HOST_SHADOW_MAX:
        dev_t._at_host = host_t
    return dev_t


def host_copy(t: torch.Tensor):
    """CPU twin of ``t`` if it is known without a synchronisation (``t`` itself when it lives on
    the CPU), else None."""
    if not t.is_cuda:
        return t
    return getattr(t, "_at_host", None)


def host_values(t: torch.Tensor) -> torch.Tensor:
    """CPU values of ``t``; synchronises only when no host twin is attached."""
    h = host_copy(t)
    return t.cpu() if h is None else h


def ensure_tensor(x, ndim: int = None, batch_size: int = None) -> torch.Tensor:
    """Make ``x`` a tensor; right-pad its shape with singleton axes up to
    ``ndim``; broadcast axis 0 to ``batch_size`` (a view, as in the reference)."""
    t = x if torch.is_tensor(x) else torch.as_tensor(x)
    h = getattr(t, "_at_host", None)
    if ndim is not None:
        assert t.ndim <= ndim
        if t.ndim < ndim:
            shape = tuple(t.shape) + (1,) * (ndim - t.ndim)
            t = t.reshape(shape)
            h = None if h is None else h.reshape(shape)
    if batch_size is not None and t.shape[0] != batch_size:
        t = t.expand(batch_size, *t.shape[1:])
        h = None if h is None else h.expand(batch_size, *h.shape[1:])
    if h is not None:
        attach_host(t, h)
    return t


def _get_value(other):
    """Operand of AudioSignal arithmetic: another signal contributes its samples."""
    from .signal import AudioSignal

    return other.audio_data if isinstance(other, AudioSignal) else other


def hz_to_bin(hz: torch.Tensor, n_fft: int, sample_rate: int) -> torch.Tensor:
    """Index of the frequency-grid point nearest to each entry of ``hz`` (core/util.py:100-126).
    The grid is the reference's: ``n_fft // 2 + 2`` points from 0 to Nyquist; frequencies above
    Nyquist are clamped to it -- on the flattened VIEW of ``hz``, i.e. in the caller's tensor when
    it is contiguous, as upstream."""
    nyquist = sample_rate / 2
    flat = hz.flatten()
    flat[flat > nyquist] = nyquist
    grid = torch.linspace(0, nyquist, n_fft // 2 + 2)
    nearest = (grid[:, None] - flat[None, :]).abs().argmin(dim=0)
    return nearest.reshape(hz.shape)


def random_state(seed: typing.Union[int, np.random.RandomState, None]):
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (numbers.Integral, np.integer, int)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("%r cannot be used to seed a numpy.random.RandomState instance" % seed)


def seed(random_seed: int, set_cudnn: bool = False):
    torch.manual_seed(random_seed)
    np.random.seed(random_seed)
    random.seed(random_seed)
    if set_cudnn:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


@contextmanager
def _close_temp_files(tmpfiles: list):
    def _close():
        for t in tmpfiles:
            t.close()
    try:
        yield
    finally:
        _close()


def sample_from_dist(dist_tuple: tuple, state: np.random.RandomState = None):
    """Draw from ("const", v) | ("uniform", lo, hi) | ("normal", mu, sd) |
    ("choice", [...]) with a seeded RandomState (transform hyper-parameters)."""
    if dist_tuple[0] == "const":
        return dist_tuple[1]
    state = random_state(state)
    fn = getattr(state, dist_tuple[0])
    return fn(*dist_tuple[1:])


def collate(list_of_dicts: list, n_splits: int = None):
    """Merge per-item (possibly nested) dicts into one batched dict:
    AudioSignals are batched with padding, tensors/arrays/scalars go through
    torch's default collate; optionally split into ``n_splits`` chunks."""
    from .signal import AudioSignal

    def flat(d, pre=()):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict) and v:
                out.update(flat(v, pre + (k,)))
            else:
                out[pre + (k,)] = v
        return out

    def unflat(d):
        out = {}
        for ks, v in d.items():
            cur = out
            for k in ks[:-1]:
                cur = cur.setdefault(k, {})
            cur[ks[-1]] = v
        return out

    batches = []
    list_len = len(list_of_dicts)
    return_list = n_splits is not None
    n_splits = 1 if n_splits is None else n_splits
    n_items = int(np.ceil(list_len / n_splits))
    for i in range(0, list_len, n_items):
        items = [flat(d) for d in list_of_dicts[i: i + n_items]]
        merged = {k: [d[k] for d in items] for k in items[0]}
        batch = {}
        for k, v in merged.items():
            if isinstance(v, list):
                if all(isinstance(s, AudioSignal) for s in v):
                    batch[k] = AudioSignal.batch(v, pad_signals=True)
                else:
                    batch[k] = torch.utils.data._utils.collate.default_collate(v)
        batches.append(unflat(batch))
    return batches if return_list else batches[0]


def prepare_batch(batch, device="cpu"):
    """Move every tensor / AudioSignal of a (nested) batch to ``device``."""
    if isinstance(batch, dict):
        return {k: prepare_batch(v, device) for k, v in batch.items()}
    if isinstance(batch, list):
        return [prepare_batch(v, device) for v in batch]
    if torch.is_tensor(batch):
        return attach_host(batch.to(device), batch)
    if hasattr(batch, "to") and hasattr(batch, "audio_data"):
        return batch.to(device)
    return batch


# ------------------------------------------------------------------ audio sources
AUDIO_EXTENSIONS = [".wav", ".flac", ".mp3", ".mp4"]
MEMORY_PREFIX = "mem://"
_memory_audio = {}


@dataclasses.dataclass
class Info:
    """Length and rate of an audio source (reference core/util.py:20-30)."""
    sample_rate: float
    num_frames: int

    @property
    def duration(self) -> float:
        return self.num_frames / self.sample_rate


def register_memory_audio(name: str, audio: torch.Tensor, sample_rate: int) -> str:
    """Make an in-memory recording addressable like a file: returns the path ``"mem://name"`` that
    ``AudioSignal(path, offset=, duration=)``, ``AudioSignal.excerpt``, ``salient_excerpt`` and
    ``AudioLoader`` accept.  ``audio`` is (C, T) or (T,), on ANY device: decoded audio that already
    sits in HBM is excerpted by slicing, with no host round trip (file decoding itself is outside
    the accelerated path)."""
    a = audio if torch.is_tensor(audio) else torch.as_tensor(audio)
    if a.ndim == 1:
        a = a[None]
    assert a.ndim == 2, "memory audio is (channels, samples)"
    if a.dtype == torch.double:
        a = a.float()
    path = MEMORY_PREFIX + name
    _memory_audio[path] = (a, int(sample_rate))
    return path


def memory_audio(path):
    return _memory_audio.get(str(path))


def info(audio_path) -> Info:
    """Sample rate and length of a source without decoding it (core/util.py:33-53)."""
    mem = memory_audio(audio_path)
    if mem is not None:
        return Info(sample_rate=mem[1], num_frames=int(mem[0].shape[-1]))
    try:
        import soundfile
    except ImportError as e:
        raise RuntimeError("reading audio files needs the optional `soundfile` package") from e
    i = soundfile.info(str(audio_path))
    return Info(sample_rate=i.samplerate, num_frames=i.frames)


def find_audio(folder: str, ext: typing.List[str] = AUDIO_EXTENSIONS):
    """Audio files below ``folder`` (behaviour of core/util.py:218-250): a path that itself ends in one of the
    extensions is returned as is (or expanded when it holds a ``*`` pattern); a directory is walked once and every
    file whose name ends in one of ``ext`` is kept, grouped by extension in the order of ``ext``."""
    text = str(folder)
    suffixes = tuple(ext)
    if text.endswith(suffixes):
        return glob.glob(text, recursive="**" in text) if "*" in text else [Path(text)]
    by_ext = {e: [] for e in ext}
    for root, _dirs, names in os.walk(text):
        for name in names:
            for e in ext:
                if name.endswith(e):
                    by_ext[e].append(Path(root) / name)
    return [f for e in ext for f in by_ext[e]]


def read_sources(sources: typing.List[str], remove_empty: bool = True, relative_path: str = "",
                 ext: typing.List[str] = AUDIO_EXTENSIONS):
    """Sources -> list of lists of ``{"path": ...}`` rows (core/util.py:253-297): a source is a
    CSV with a ``path`` column, a folder of audio files, or -- for audio that is already decoded --
    a list of ``mem://`` paths / row dicts."""
    files = []
    relative_path = Path(relative_path)
    for source in sources:
        _files = []
        if isinstance(source, (list, tuple)):
            for x in source:
                _files.append(dict(x) if isinstance(x, dict) else {"path": str(x)})
            files.append(sorted(_files, key=lambda x: x["path"]))
            continue
        source = str(source)
        if source.endswith(".csv"):
            with open(source, "r") as f:
                for x in csv.DictReader(f):
                    if remove_empty and x["path"] == "":
                        continue
                    if x["path"] != "":
                        x["path"] = str(relative_path / x["path"])
                    _files.append(x)
        else:
            for x in find_audio(source, ext=ext):
                _files.append({"path": str(relative_path / x)})
        files.append(sorted(_files, key=lambda x: x["path"]))
    return files


def choose_from_list_of_lists(state: np.random.RandomState, list_of_lists: list, p: float = None):
    """One item of a list of lists: the list by ``p``, the item uniformly (core/util.py:300-322)."""
    source_idx = state.choice(list(range(len(list_of_lists))), p=p)
    item_idx = state.randint(len(list_of_lists[source_idx]))
    return list_of_lists[source_idx][item_idx], source_idx, item_idx


class chdir:
    """``with util.chdir(path):`` -- run the block with ``path`` as the working directory, restore the previous one
    on exit, also when the block raises (core/util.py:325-343)."""

    def __init__(self, newdir):
        self._target = newdir
        self._stack = []

    def __enter__(self):
        self._stack.append(os.getcwd())
        os.chdir(self._target)
        return self

    def __exit__(self, *exc):
        os.chdir(self._stack.pop())
        return False


_JOINERS = {"tuple": None, "dot": ".", "underscore": "_", "path": "/"}


def flatten(d: dict, reducer="tuple", keep_empty_types=()) -> dict:
    """Nested dictionary -> flat dictionary (the ``flatten_dict`` package the reference's util imports): keys are tuples
    of the path (``reducer="tuple"``, the default the reference uses) or the path joined with "." / "_" / "/"
    (``"dot"``, ``"underscore"``, ``"path"``) or passed through a callable ``reducer(parent_key, key)``.  Empty
    dictionaries are dropped, as flatten_dict does by default; ``keep_empty_types=(dict,)`` keeps them as leaves.
    With the string reducers, keys are joined through ``str``: non-string keys do not round-trip (5 comes back as "5")."""
    from .transforms import _flatten
    flat = _flatten(d, keep_empty=dict in tuple(keep_empty_types))
    if callable(reducer):
        out = {}
        for ks, v in flat.items():
            key = None
            for k in ks:
                key = reducer(key, k)
            out[key] = v
        return out
    if reducer not in _JOINERS:
        raise ValueError(f"unknown reducer {reducer!r}")
    sep = _JOINERS[reducer]
    return flat if sep is None else {sep.join(str(k) for k in ks): v for ks, v in flat.items()}


def unflatten(d: dict, splitter="tuple") -> dict:
    """Inverse of :func:`flatten` for the same ``splitter`` names (or a callable key -> tuple of path elements)."""
    from .transforms import _unflatten
    if callable(splitter):
        return _unflatten({tuple(splitter(k)): v for k, v in d.items()})
    if splitter not in _JOINERS:
        raise ValueError(f"unknown splitter {splitter!r}")
    sep = _JOINERS[splitter]
    return _unflatten(d if sep is None else {tuple(str(k).split(sep)): v for k, v in d.items()})
