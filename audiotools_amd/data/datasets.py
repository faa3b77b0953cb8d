"""``AudioLoader``: excerpts from lists of audio sources, a BATCH at a time.

Drop-in for the reference's ``audiotools/data/datasets.py:15-136`` -- same constructor, same
``__call__(state, sample_rate, duration, ...)`` item dictionary, the same draws from every caller's
``RandomState`` -- built the other way round: the unit of work is ``batch(states, ...)``, B items at once,
and ``__call__`` is the B = 1 case of it.

    plan      host only, per item, in the reference's draw order: pick (source, item) with
              ``state.choice`` / ``state.randint``; then the excerpt position -- a fixed ``offset``, ONE
              ``state.uniform`` (no loudness cutoff), or, with a cutoff, up to ``num_tries`` candidate
              positions drawn from a COPY of the state;
    measure   all candidates of all items of one source layout (rate, channels, device) are stacked from
              slices of the registered recordings and measured by ONE batched ``loudness()`` (the native LUFS
              kernel when the recordings live in HBM), ONE synchronisation; the first candidate above the
              cutoff wins and the item's real state is advanced by exactly the draws the reference's
              try-until-loud loop would have made;
    gather    the winning excerpts are stacked per layout, mixed down, resampled (one polyphase launch per
              distinct source rate) and zero-padded to the requested duration; the result is one
              (B, C, duration * sample_rate) signal that never visited the host.

Items that cannot take the batched route -- sources decoded from files (optional ``soundfile``; decoding is
outside the accelerated path), or an excerpt that runs past the end of a short recording (the
reference resamples the SHORT excerpt and pads afterwards, which is not the same as padding first) -- go
through the one-item route with the same draws.  ``tests/test_feeding.py`` pins both routes seed for seed
to the unmodified reference.

The dataset classes around it (AudioDataset, ConcatDataset, samplers) are training-loop plumbing and stay out
of scope.
"""
from typing import Callable, List

import numpy as np
import torch

from .. import util
from ..signal import AudioSignal

NUM_TRIES = 8          # AudioSignal.salient_excerpt's default (audio_signal.py:232)


def default_matcher(x, y) -> bool:
    """Two paths belong to the same multitrack item when they sit in the same directory (datasets.py:138-139)."""
    from pathlib import Path
    return Path(x).parent == Path(y).parent


def align_lists(lists, matcher: Callable = default_matcher):
    """Line up several source lists (one per stem of a multitrack collection) item by item: walking the longest list,
    every other list gets a ``{"path": "none"}`` placeholder wherever its entry at that position does not belong to the
    same item, or where it has run out (datasets.py:142-150).  In place, like the reference; returns ``lists``."""
    anchor = max(lists, key=len) if lists else []
    pos = 0
    while pos < len(anchor):
        want = anchor[pos]["path"]
        for other in lists:
            if pos >= len(other):
                other.append({"path": "none"})
            elif not matcher(other[pos]["path"], want):
                other.insert(pos, {"path": "none"})
        pos += 1
    return lists


def _probe_uniform(state, lo, hi, n):
    """The next ``n`` values ``state.uniform(lo, hi)`` WOULD return, without consuming them."""
    twin = np.random.RandomState()
    twin.set_state(state.get_state())
    return [twin.uniform(lo, hi) for _ in range(n)]


class _Plan:
    """What one item needs: where its audio comes from and where the excerpt starts."""
    __slots__ = ("row", "source_idx", "item_idx", "path", "mem", "start", "offset_s", "n_src", "candidates", "bounds", "one_item")

    def __init__(self, row, source_idx, item_idx):
        self.row, self.source_idx, self.item_idx = row, source_idx, item_idx
        self.path = row["path"]
        self.mem = None if self.path == "none" else util.memory_audio(self.path)
        self.start = None           # first source sample of the excerpt
        self.offset_s = None        # the same in seconds, exactly as drawn (metadata["offset"])
        self.n_src = None           # excerpt length in source samples
        self.candidates = None      # candidate offsets (seconds) still to be measured
        self.bounds = None          # (lo, hi) of the uniform draw
        self.one_item = False       # take the one-item route


class AudioLoader:
    """Loads audio endlessly from a list of audio sources.

    sources        list of CSV paths / folders, or -- for decoded audio -- lists of ``mem://`` paths or row dicts
                   ``{"path": "mem://...", ...extra columns...}``
    weights        probability of each source
    transform      optional transform instantiated alongside every item
    shuffle        shuffle the (source, item) index used by ``global_idx`` lookups (seeded by ``shuffle_state``)
    """

    def __init__(self, sources: List[str] = None, weights: List[float] = None, transform: Callable = None,
                 relative_path: str = "", ext: List[str] = util.AUDIO_EXTENSIONS, shuffle: bool = True,
                 shuffle_state: int = 0):
        self.sources, self.weights, self.transform = sources, weights, transform
        self.audio_lists = util.read_sources(sources, relative_path=relative_path, ext=ext)
        order = [(s, i) for s, rows in enumerate(self.audio_lists) for i in range(len(rows))]
        if shuffle:
            util.random_state(shuffle_state).shuffle(order)
        self.audio_indices = order

    # ------------------------------------------------------------------ plan
    def _row(self, s, i):
        lists = self.audio_lists
        if -len(lists) <= s < len(lists) and -len(lists[s]) <= i < len(lists[s]):
            return lists[s][i]
        return {"path": "none"}                    # an index past the lists reads as silence (datasets.py:83-86)

    def _pick(self, state, source_idx, item_idx, global_idx):
        """(row, source_idx, item_idx) of one item; only the free choice draws from ``state``."""
        if source_idx is not None and item_idx is not None:
            return self._row(source_idx, item_idx), source_idx, item_idx
        if global_idx is not None:
            source_idx, item_idx = self.audio_indices[global_idx % len(self.audio_indices)]
            return self.audio_lists[source_idx][item_idx], source_idx, item_idx
        return util.choose_from_list_of_lists(state, self.audio_lists, p=self.weights)

    def _plan(self, state, duration, loudness_cutoff, offset, source_idx, item_idx, global_idx):
        plan = _Plan(*self._pick(state, source_idx, item_idx, global_idx))
        if plan.path == "none":
            return plan
        if plan.mem is None:                       # a file: decoded by the one-item route
            plan.one_item = True
            plan.offset_s = offset
            return plan
        bank, sr = plan.mem
        plan.n_src = int(duration * sr)
        if offset is not None:
            plan.offset_s = offset
        else:
            plan.bounds = (0, max(bank.shape[-1] / sr - duration, 0))
            if loudness_cutoff is None:
                plan.offset_s = state.uniform(*plan.bounds)               # AudioSignal.excerpt's one draw
            else:
                plan.candidates = _probe_uniform(state, *plan.bounds, NUM_TRIES)
        if plan.offset_s is not None:
            plan.start = int(plan.offset_s * sr)
        starts = [plan.start] if plan.candidates is None else [int(o * sr) for o in plan.candidates]
        # an excerpt that runs past the end of a short recording: the reference measures / resamples the SHORT
        # signal and pads afterwards -- the one-item route (candidates were only probed: the state is untouched)
        plan.one_item = any(s + plan.n_src > bank.shape[-1] for s in starts)
        return plan

    # --------------------------------------------------------------- one item
    def _one_item(self, plan, state, sample_rate, duration, loudness_cutoff, num_channels):
        """The sequential route for one item (files; excerpts past the end of a short recording)."""
        if plan.offset_s is not None:              # fixed offset, or the single draw already taken while planning
            sig = AudioSignal(plan.path, offset=plan.offset_s, duration=duration)
        else:
            sig = AudioSignal.salient_excerpt(plan.path, duration=duration, state=state, loudness_cutoff=loudness_cutoff)
        if num_channels == 1:
            sig = sig.to_mono()
        sig = sig.resample(sample_rate)
        if sig.duration < duration:
            sig = sig.zero_pad_to(int(duration * sample_rate))
        return sig

    # ------------------------------------------------------------------ batch
    def batch(self, states, sample_rate: int, duration: float, loudness_cutoff: float = -40, num_channels: int = 1,
              offset: float = None, source_idx: int = None, item_idx: int = None, global_idx: int = None,
              as_list: bool = False):
        """B items at once.  ``states``: one seed / ``RandomState`` per item (each is consumed exactly as
        one ``__call__`` would consume it).  Returns ``{"signal": (B, C, T) AudioSignal, "source_idx": [...],
        "item_idx": [...], "source": [...], "path": [...]}``; with ``as_list`` the signals stay a list of B
        one-item signals (items of different channel counts cannot share a tensor)."""
        rs = [util.random_state(s) for s in states]
        B = len(rs)
        n_out = int(duration * sample_rate)
        plans, singles = [], {}
        for b, st in enumerate(rs):
            plan = self._plan(st, duration, loudness_cutoff, offset, source_idx, item_idx, global_idx)
            plans.append(plan)
            if plan.one_item:
                singles[b] = self._one_item(plan, st, sample_rate, duration, loudness_cutoff, num_channels)

        # ---- measure: candidates of every item that has them, one stacked loudness() per source layout
        groups = {}
        for b, plan in enumerate(plans):
            if plan.one_item or plan.mem is None or plan.candidates is None:
                continue
            bank, sr = plan.mem
            groups.setdefault((sr, bank.shape[0], bank.device, plan.n_src), []).append(b)
        for (sr, _c, _dev, n_src), members in groups.items():
            rows = []
            for b in members:
                bank = plans[b].mem[0]
                rows += [bank[:, int(o * sr): int(o * sr) + n_src] for o in plans[b].candidates]
            loud = AudioSignal(torch.stack(rows), sr).loudness().reshape(len(members), NUM_TRIES)
            above = (loud > loudness_cutoff)
            first = torch.where(above.any(1), above.float().argmax(1), torch.full((len(members),), NUM_TRIES - 1, device=loud.device))
            first = first.cpu().tolist()                                   # the one synchronisation
            for b, k in zip(members, first):
                plan = plans[b]
                for _ in range(k + 1):                                     # the draws the try-until-loud loop makes
                    rs[b].uniform(*plan.bounds)
                plan.offset_s = plan.candidates[k]
                plan.start = int(plan.offset_s * sr)

        # ---- gather: winners stacked per source layout, mixed down, resampled, padded
        out = [None] * B
        layouts = {}
        for b, plan in enumerate(plans):
            if b in singles:
                out[b] = singles[b]
            elif plan.mem is None:                                         # path "none": silence
                out[b] = AudioSignal.zeros(duration, sample_rate, num_channels)
            else:
                bank, sr = plan.mem
                layouts.setdefault((sr, bank.shape[0], bank.device, plan.n_src), []).append(b)
        whole = None
        for (sr, _c, _dev, n_src), members in layouts.items():
            sig = AudioSignal(torch.stack([plans[b].mem[0][:, plans[b].start: plans[b].start + n_src] for b in members]), sr)
            if num_channels == 1:
                sig = sig.to_mono()
            sig = sig.resample(sample_rate)
            if sig.signal_length < n_out:
                sig = sig.zero_pad_to(n_out)
            if len(members) == B and not as_list:
                whole = sig                                                # one layout covers the batch: no re-stacking
            else:
                for j, b in enumerate(members):
                    out[b] = sig[j]
        # per-item metadata, the same on both routes: what AudioSignal(path, offset, duration) records, then the row's
        # columns -- which win, as in the reference (datasets.py:117-124 writes them after the load)
        metas = []
        for b, plan in enumerate(plans):
            m = {}
            if plan.mem is not None and b not in singles:
                m["offset"], m["duration"] = plan.offset_s, duration
            m.update(plan.row)
            metas.append(m)
        if whole is None:
            for b in range(B):
                out[b].metadata.update(metas[b])
            whole = out if as_list else (AudioSignal.batch(out, pad_signals=True) if B > 1 else out[0])

        item = {"signal": whole, "metadata": metas,
                "source_idx": [p.source_idx for p in plans], "item_idx": [p.item_idx for p in plans],
                "source": [str(self.sources[p.source_idx]) for p in plans], "path": [str(p.path) for p in plans]}
        if self.transform is not None:
            sigs = out if out[0] is not None else [whole[j] for j in range(B)]
            item["transform_args"] = [self.transform.instantiate(st, signal=s) for st, s in zip(rs, sigs)]
        return item

    def __call__(self, state, sample_rate: int, duration: float, loudness_cutoff: float = -40, num_channels: int = 1,
                 offset: float = None, source_idx: int = None, item_idx: int = None, global_idx: int = None):
        """One item: the B = 1 case of :meth:`batch`, unwrapped to the reference's item dictionary
        (datasets.py:126-136)."""
        got = self.batch([state], sample_rate, duration, loudness_cutoff=loudness_cutoff, num_channels=num_channels,
                         offset=offset, source_idx=source_idx, item_idx=item_idx, global_idx=global_idx, as_list=True)
        item = {k: v[0] for k, v in got.items() if k != "metadata"}       # (the signal carries it: the reference's item keys)
        return item
