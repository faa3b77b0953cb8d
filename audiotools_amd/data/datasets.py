"""``AudioLoader``: draws excerpts from lists of audio sources (reference
``audiotools/data/datasets.py:15-136``), with the same call signature, the same draws from the
caller's ``RandomState`` and the same post-processing (``to_mono`` -> ``resample`` ->
``zero_pad_to``), so transforms that own a loader (BackgroundNoise, CrossTalk,
RoomImpulseResponse) and dataset code written against the reference work unchanged.

What is different is WHERE the audio lives.  Sources may be CSV files / folders (decoded through
the optional ``soundfile`` package: file decoding is outside the accelerated path), or recordings
that are already decoded and registered with ``util.register_memory_audio`` -- on the host or in
HBM.  For HBM-resident sources the whole ``__call__`` runs on the device: the excerpt is a slice,
``salient_excerpt`` measures all its candidate windows with one batched LUFS launch,
``resample`` is the polyphase kernel, and the result never visits the host.

The dataset classes around it (AudioDataset, ConcatDataset, samplers) are training-loop plumbing
and stay out of scope.
"""
from typing import Callable, List

from .. import util
from ..signal import AudioSignal


class AudioLoader:
    """Loads audio endlessly from a list of audio sources (datasets.py:15-68).

    sources        list of CSV paths / folders, or -- for decoded audio -- lists of ``mem://``
                   paths or row dicts ``{"path": "mem://...", ...extra columns...}``
    weights        probability of each source
    transform      optional transform instantiated alongside every item
    shuffle        shuffle the (source, item) index used by ``global_idx`` lookups
    """

    def __init__(self, sources: List[str] = None, weights: List[float] = None, transform: Callable = None,
                 relative_path: str = "", ext: List[str] = util.AUDIO_EXTENSIONS, shuffle: bool = True,
                 shuffle_state: int = 0):
        self.audio_lists = util.read_sources(sources, relative_path=relative_path, ext=ext)
        self.audio_indices = [(src_idx, item_idx) for src_idx, src in enumerate(self.audio_lists)
                              for item_idx in range(len(src))]
        if shuffle:
            state = util.random_state(shuffle_state)
            state.shuffle(self.audio_indices)
        self.sources = sources
        self.weights = weights
        self.transform = transform

    def __call__(self, state, sample_rate: int, duration: float, loudness_cutoff: float = -40,
                 num_channels: int = 1, offset: float = None, source_idx: int = None, item_idx: int = None,
                 global_idx: int = None):
        if source_idx is not None and item_idx is not None:
            try:
                audio_info = self.audio_lists[source_idx][item_idx]
            except Exception:
                audio_info = {"path": "none"}
        elif global_idx is not None:
            source_idx, item_idx = self.audio_indices[global_idx % len(self.audio_indices)]
            audio_info = self.audio_lists[source_idx][item_idx]
        else:
            audio_info, source_idx, item_idx = util.choose_from_list_of_lists(state, self.audio_lists, p=self.weights)

        path = audio_info["path"]
        signal = AudioSignal.zeros(duration, sample_rate, num_channels)
        if path != "none":
            if offset is None:
                signal = AudioSignal.salient_excerpt(path, duration=duration, state=state, loudness_cutoff=loudness_cutoff)
            else:
                signal = AudioSignal(path, offset=offset, duration=duration)

        if num_channels == 1:
            signal = signal.to_mono()
        signal = signal.resample(sample_rate)
        if signal.duration < duration:
            signal = signal.zero_pad_to(int(duration * sample_rate))

        for k, v in audio_info.items():
            signal.metadata[k] = v

        item = {"signal": signal, "source_idx": source_idx, "item_idx": item_idx,
                "source": str(self.sources[source_idx]), "path": str(path)}
        if self.transform is not None:
            item["transform_args"] = self.transform.instantiate(state, signal=signal)
        return item
