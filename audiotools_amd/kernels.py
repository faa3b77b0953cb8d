"""Thin Python launchers for the ``at_*`` C-ABI entry points.

Each function takes torch tensors that already live on a HIP device,
allocates the outputs/workspaces with ``torch.empty`` (torch owns all memory,
SURVEY.md 8(b) "Ownership"), and launches on torch's current stream.  No
function here computes anything on the host beyond shapes.
"""
import functools
import math
import threading
import os

import numpy as np

import torch

from . import _native, tables

# Development A/B switches exist only when the process opted in with AT_DEV_KNOBS=1 (the tools under tools/ do, together
# with a -DAT_DEV_KNOBS=1 build of the library): the shipped path never changes with the caller's environment.
_DEV_KNOBS = os.environ.get("AT_DEV_KNOBS") == "1"


def _knob(name, default):
    return os.environ.get(name, default) if _DEV_KNOBS else default

PAD_MODES = {"reflect": 0, "constant": 1, "replicate": 2, "circular": 3}


def is_native(t: torch.Tensor) -> bool:
    """Dispatch rule: HIP tensor, float32, no autograd."""
    return t.is_cuda and t.dtype == torch.float32 and not (t.requires_grad and torch.is_grad_enabled())


def _require_native_ok(t):
    if not t.is_cuda:
        raise _native.NativeError("native kernels need a HIP device tensor")


def stft_frames(T: int, n_fft: int, hop: int, pad: int, right_pad: int, match_stride: bool):
    """(first computed frame, number of output frames) of the reference's
    stft(): torch.stft(center=True) on the outer-padded signal gives
    1 + T2//hop frames; match_stride drops 2 on each side
    (audio_signal.py:1203-1209)."""
    T2 = T + 2 * pad + right_pad
    n_total = 1 + T2 // hop
    if match_stride:
        return 2, max(n_total - 4, 0)
    return 0, n_total


def stft_native_supported(n_fft: int) -> bool:
    """Some native kernel covers this transform size (fused wave-FFT kernels or the generic
    mixed-radix one)."""
    return bool(_native.lib().at_stft_native_supported(int(n_fft)))


def stft_fused_supported(n_fft: int) -> bool:
    """The fused wave-FFT kernels cover it (powers of two up to 2048): fused mel, adjoint kernels."""
    return bool(_native.lib().at_stft_fused_supported(int(n_fft)))


# ---------------------------------------------------------------------------------------------------------------------
# Placement-aware output buffers (round 5; an OPT-IN since round 6).  The time of the fused STFT kernel follows WHERE its
# spectrum buffer lies physically: the same binary writes the same 7.2 GB in 1.95 ms into one allocation and in 2.12-2.19 ms
# into the next one of the same process (DRAM bank-level parallelism behind GB-scale address bits; profiles/r05_notes.md
# section 1) -- deterministic per allocation, invisible from the virtual address, not reachable by any schedule of the kernel.
# A process that has HBM to spare can ask the library to CHOOSE (`output_placement(enabled=True)`): once a large shape has
# been seen CALIBRATE_AFTER times, the call allocates a few candidate buffers, times the REAL kernel into each (every
# candidate holds this call's valid result), keeps the fastest KEEP in a small pool and recycles them for later calls of the
# shape.  Only the SPECTRUM buffer (inverse: the signal buffer) is pooled: exchanging buffers between a slow and a fast set
# showed that the other operands' placement does not matter (r05_notes.md 1.1).  Semantics are those of a fresh allocation:
# a pooled buffer is handed out again only when no tensor of a caller references its storage any more (the C++ storage use
# count; the caller's tensor object is created under the pool's lock), otherwise the call falls back to torch.empty.
#
# What it costs, and the limits that keep it a good citizen:
#   * OFF unless the process opts in; nothing is calibrated before the CALIBRATE_AFTER-th call of a shape (variable-length
#     workloads never calibrate), at most MAX_CALIBRATIONS shapes per process;
#   * calibration takes at most FREE_FRACTION of the memory that is free at that moment (candidates that lose go back to
#     torch's caching allocator), runs 3 launches per candidate and ONE device synchronisation;
#   * KEEP buffers per shape stay pinned (default 1), all shapes together at most MAX_POOL_BYTES; `release_workspaces()`
#     (also called when an allocation of this module hits OutOfMemoryError, before one retry) drops them;
#   * not honoured: `Tensor.record_stream` by a consumer on another stream (the pool is keyed per stream and reuses a
#     buffer as soon as the last reference is gone -- keep a reference until the other stream's work is enqueued-complete,
#     or leave the pool off), and a holder that keeps ONLY an `UntypedStorage` Python object of a result (torch re-uses
#     the storage's preserved PyObject: the use count does not see it).
class _PlacedOutputs:
    CANDIDATES = 12             # buffers timed at calibration (fewer when FREE_FRACTION of the free memory does not hold them).  The times
                                # of one process spread over about 12 %; the fastest of eight was 1.864-1.940 ms in nine processes
                                # (profiles/r05_notes.md 5): four more draws for ~30 ms more calibration, once per shape
    KEEP = 1                    # ... of which this many stay pinned (AudioSignal.stft / mel_spectrogram release the spectrum they replace
                                # before the kernel runs: one buffer serves a signal's repeated transforms; callers that hold several results
                                # of a shape at once may ask for more)
    MIN_BYTES = 256 << 20       # outputs below this size are not worth it
    MAX_POOL_BYTES = 24 << 30   # all shapes together; the least recently used shape is dropped beyond it
    CALIBRATE_AFTER = 3         # a shape is calibrated at its N-th call (the earlier ones take plain allocations)
    MAX_CALIBRATIONS = 4        # calibrations per process (each is ~37 launches and a device synchronisation)
    FREE_FRACTION = 0.5         # of the memory free at calibration time
    enabled = False             # opt-in: output_placement(enabled=True)
    available = hasattr(torch._C, "_storage_Use_Count")      # (a private torch API: without it nothing is pooled)

    def __init__(self):
        self.shapes = {}        # key -> {"slots": [(ms, buf)], "bytes": int, "tick": int} | int (calls seen so far) | None (not worth it / no room)
        self.tick = 0
        self.calibrations = 0
        self.lock = threading.RLock()      # two host threads calling the same shape: one calibrates, the other waits for the pool

    @staticmethod
    def _free(t):
        # 2 = the pool's tensor + the temporary storage object of this query; every alias / view / saved tensor / numpy export adds one
        return t is None or torch._C._storage_Use_Count(t.untyped_storage()._cdata) <= 2

    def acquire(self, key, nbytes, alloc, launch):
        """(buf, launched) for this call: an ALIAS (created under the lock, so that a second host thread sees the buffer as
        held before this one has launched anything) of a pooled buffer when the shape has a calibrated pool and one of its
        buffers is free; at the CALIBRATE_AFTER-th call of a shape, calibration (then `launched` is True: the buffer already
        holds this call's result); otherwise None (plain allocation)."""
        if not (self.enabled and self.available) or nbytes < self.MIN_BYTES or torch.cuda.is_current_stream_capturing():
            return None
        with self.lock:
            self.tick += 1
            ent = self.shapes.get(key, 0)
            if ent is None:
                return None
            if isinstance(ent, int):
                seen = ent + 1
                if seen < self.CALIBRATE_AFTER:
                    self.shapes[key] = seen
                    if len(self.shapes) > 256:      # a workload of ever-changing shapes: forget the counters, keep the pools
                        self.shapes = {k: e for k, e in self.shapes.items() if isinstance(e, dict)}
                    return None
                if self.calibrations >= self.MAX_CALIBRATIONS:
                    self.shapes[key] = None
                    return None
                try:
                    got = self._calibrate(key, nbytes, alloc, launch)
                except torch.cuda.OutOfMemoryError:    # somebody else took the memory between the check and the allocations
                    self.shapes[key] = None
                    return None
                return None if got is None else (got[0].detach(), got[1])
            ent["tick"] = self.tick
            for _ms, sb in ent["slots"]:
                if self._free(sb):
                    return sb.detach(), False
            return None

    def _calibrate(self, key, nbytes, alloc, launch):
        dev = key[0]
        free_b, _total = torch.cuda.mem_get_info(dev)
        n_cand = min(self.CANDIDATES, int(self.FREE_FRACTION * free_b // nbytes))
        if n_cand < 2 or self.KEEP * nbytes > self.MAX_POOL_BYTES:
            self.shapes[key] = None                # no room for candidates, or a shape the pool is not meant to hold
            return None
        while sum(e["bytes"] for e in self.shapes.values() if isinstance(e, dict)) + self.KEEP * nbytes > self.MAX_POOL_BYTES:
            live = [(e["tick"], k) for k, e in self.shapes.items() if isinstance(e, dict)]
            if not live:
                break
            del self.shapes[min(live)[1]]
        self.calibrations += 1
        cands = [alloc() for _ in range(n_cand)]
        for sb in cands:                           # first touch of every buffer
            launch(sb)
        evs = []
        for _round in range(2):                    # two timed launches per buffer, interleaved over them; the smaller one counts
            for sb in cands:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(sb)
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize(dev)                # once per shape and process: calibration, not the transform path
        n = len(cands)
        timed = sorted((min(evs[i][0].elapsed_time(evs[i][1]), evs[n + i][0].elapsed_time(evs[n + i][1])), i) for i in range(n))
        slots = [(ms, cands[i]) for ms, i in timed[: self.KEEP]]
        self.shapes[key] = {"slots": slots, "bytes": len(slots) * nbytes, "tick": self.tick,
                            "calibration_ms": [ms for ms, _ in timed]}
        return slots[0][1], True

    def release(self):
        """Drop every pooled buffer (they return to torch's caching allocator) and the per-shape call counters."""
        with self.lock:
            self.shapes.clear()

    def bytes_held(self):
        with self.lock:
            return sum(e["bytes"] for e in self.shapes.values() if isinstance(e, dict))

    def report(self):
        with self.lock:
            return [{"op": k[3] if len(k) > 3 else "stft", "shape": list(k[1]), "calibration_ms": e["calibration_ms"],
                     "kept_ms": [ms for ms, _ in e["slots"]], "bytes_held": e["bytes"]}
                    for k, e in self.shapes.items() if isinstance(e, dict)]


_placed_outputs = _PlacedOutputs()


def output_placement(enabled: bool = None, candidates: int = None, keep: int = None, min_bytes: int = None,
                     calibrate_after: int = None, max_pool_bytes: int = None):
    """Configure / query the placement-aware pool of large STFT outputs (see _PlacedOutputs; OFF by default).  Returns its
    report: per calibrated shape the kernel times measured on the candidate buffers, on the ones that were kept, and the
    bytes they pin.  Disabling drops the pooled buffers (as does `release_workspaces()`)."""
    if enabled is not None:
        _PlacedOutputs.enabled = bool(enabled) and _PlacedOutputs.available
        if not enabled:
            _placed_outputs.release()
    if candidates is not None:
        _PlacedOutputs.CANDIDATES = max(int(candidates), 1)
    if keep is not None:
        _PlacedOutputs.KEEP = max(int(keep), 1)
    if min_bytes is not None:
        _PlacedOutputs.MIN_BYTES = int(min_bytes)
    if calibrate_after is not None:
        _PlacedOutputs.CALIBRATE_AFTER = max(int(calibrate_after), 1)
    if max_pool_bytes is not None:
        _PlacedOutputs.MAX_POOL_BYTES = int(max_pool_bytes)
    return _placed_outputs.report()


def release_workspaces(empty_cache: bool = False) -> int:
    """Give back what this module keeps between calls: the pooled output buffers of `output_placement` and the cached
    scratch workspaces.  Returns the number of bytes the pool had pinned.  ``empty_cache``: also return torch's cached
    blocks to the driver.  Called by the launchers themselves when an allocation runs out of memory (then retried once)."""
    held = _placed_outputs.bytes_held()
    _placed_outputs.release()
    _ws_cache.clear()          # (the conv workspace of cfg4 is several GB; the next kernel call re-allocates what it needs)
    if empty_cache and torch.cuda.is_available():
        torch.cuda.empty_cache()
    return held


def _alloc_retry(alloc):
    """``alloc()``; on OutOfMemoryError the pool and the workspaces are released and the allocation is tried once more."""
    try:
        return alloc()
    except torch.cuda.OutOfMemoryError:
        release_workspaces(empty_cache=True)
        return alloc()


def stft_mel(audio: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, *, pad: int = 0,
             right_pad: int = 0, padding_type: str = "reflect", match_stride: bool = False,
             want_stft: bool = True, mel=None, frame_range=None, out=None):
    """Fused STFT (+ mel).  ``audio`` (B, C, T) float32 HIP tensor.

    ``out``: optional ``(stft_buf, mel_buf)`` -- caller-provided contiguous output buffers of the physical shapes
    (B, C, N, F) complex64 and (B, C, N, n_mels) float32 (``mel_buf`` None without ``mel``) instead of fresh allocations.

    ``mel`` is ``None`` or a tuple ``(unit_info, unit_w, n_mels)`` of
    device tables from :func:`tables.mel_units`.

    Returns ``(stft, mel_spec)``: ``stft`` is a complex64 tensor of logical
    shape (B, C, F, N) whose memory is bin-contiguous (B, C, N, F) -- the same
    physical layout torch.stft hands the reference -- and ``mel_spec`` is the
    (B, C, n_mels, N) transposed view of a (B, C, N, n_mels) buffer
    (audio_signal.py:1367-1368).
    """
    _require_native_ok(audio)
    if padding_type not in PAD_MODES:
        raise NotImplementedError(f"Unrecognised padding mode {padding_type}")
    B, C, T = audio.shape
    audio = audio.contiguous()
    F = n_fft // 2 + 1
    frame_lo, n_out = stft_frames(T, n_fft, hop, pad, right_pad, match_stride)
    if frame_range is not None:      # explicit (first frame, count) of the centre-padded transform
        frame_lo, n_out = frame_range
    dev = audio.device
    tw = tables.stft_twiddles(n_fft, dev)
    if not want_stft:
        raise NotImplementedError("the fused kernel always produces stft_data")
    info = w = None
    n_units = n_mels = 0
    if mel is not None:
        info, w, n_mels = mel
        # fused sizes: unit tables (n_units rows of 2 ints); generic sizes: banded tables (n_units = number of 16-bin chunks)
        n_units = int(info.shape[0]) if stft_fused_supported(n_fft) else int(w.shape[0])
    shape_s, shape_m = (B, C, n_out, F), ((B, C, n_out, n_mels) if mel is not None else None)

    def alloc():
        return torch.empty(shape_s, dtype=torch.complex64, device=dev)

    def launch(sb, mb):
        code = _native.lib().at_stft_mel_f32(
            _native.ptr(audio), B * C, T, _native.ptr(window), _native.ptr(tw), n_fft, hop, pad, right_pad,
            PAD_MODES[padding_type], frame_lo, n_out, _native.ptr(sb), _native.ptr(info), _native.ptr(w),
            n_units, n_mels, _native.ptr(mb), _native.current_stream(dev))
        _native.check(code, "at_stft_mel_f32")

    launched = False
    if out is not None:
        stft_buf = out[0]
        if not (isinstance(stft_buf, torch.Tensor) and stft_buf.shape == shape_s and stft_buf.dtype == torch.complex64
                and stft_buf.is_contiguous() and stft_buf.device == dev):      # (raw pointers go to the kernel: never an `assert`)
            raise ValueError("out[0]: contiguous complex64 (B, C, N, F) on the audio's device")
        mel_buf = out[1] if mel is not None else None
        if mel is not None and not (isinstance(mel_buf, torch.Tensor) and mel_buf.shape == shape_m and mel_buf.dtype == torch.float32
                                    and mel_buf.is_contiguous() and mel_buf.device == dev):
            raise ValueError("out[1]: contiguous float32 (B, C, N, n_mels) on the audio's device")
    else:
        mel_buf = _alloc_retry(lambda: torch.empty(shape_m, dtype=torch.float32, device=dev)) if shape_m else None
        nbytes = B * C * n_out * F * 8
        key = (dev, shape_s, torch.cuda.current_stream(dev).cuda_stream)      # (with or without the mel stage: one pool per spectrum shape)
        # (every calibration launch writes the same mel values into the one mel buffer of this call)
        got = _placed_outputs.acquire(key, nbytes, alloc, lambda sb: launch(sb, mel_buf)) if nbytes >= _PlacedOutputs.MIN_BYTES else None
        if got is not None:
            stft_buf, launched = got
        else:
            stft_buf = _alloc_retry(alloc)
    if not launched:
        launch(stft_buf, mel_buf)
    stft = stft_buf.transpose(2, 3) if stft_buf is not None else None
    mel_spec = mel_buf.transpose(2, 3) if mel_buf is not None else None
    return stft, mel_spec


def stft_mel_floor(audio: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, stft_buf: torch.Tensor, mel=None,
                   mel_buf: torch.Tensor = None) -> bool:
    """MEASUREMENT: one launch of the zero-compute twin of the n_fft 2048 / hop 512 kernel (at_stft_mel_floor_f32, an entry
    point of the DEVELOPMENT library only: `_native.dev_lib()`) into caller-provided buffers of the shapes ``stft_mel``
    allocates; False when that library is not built or the kernel does not take the shape."""
    _require_native_ok(audio)
    B, C, T = audio.shape
    dev = audio.device
    tw = tables.stft_twiddles(n_fft, dev)
    info = w = None
    n_units = n_mels = 0
    if mel is not None:
        info, w, n_mels = mel
        n_units = int(info.shape[0])
    dlib = _native.dev_lib()
    if dlib is None:
        return False                     # the development library is not built: no floor to measure
    code = dlib.at_stft_mel_floor_f32(
        _native.ptr(audio), B * C, T, _native.ptr(window), _native.ptr(tw), n_fft, hop, 0, 0, PAD_MODES["reflect"], 0,
        1 + T // hop, _native.ptr(stft_buf), _native.ptr(info), _native.ptr(w), n_units, n_mels, _native.ptr(mel_buf),
        _native.current_stream(dev))
    if code == -2:
        return False
    _native.check(code, "at_stft_mel_floor_f32")
    return True


def _ola_envelope(window: torch.Tensor, hop: int, n_frames: int, dtype=torch.float32) -> torch.Tensor:
    """sum_f w^2[n - f hop] over ``n_frames`` frames: the overlap-add envelope torch.istft divides by, length
    (n_frames - 1) hop + n_fft.  One index_add_ in float64 (the order of <= n_fft / hop additions per sample cannot show in
    the float32 result).  Deliberately NOT ``conv_transpose1d``: that is a MIOpen call, and the first MIOpen call of a process
    made from autograd's worker thread -- where the adjoints below run -- aborted the interpreter on this stack (round 6,
    sessions s05 / s06: a backward pass before any forward convolution)."""
    n_fft = window.shape[-1]
    w2 = window.to(torch.float64) ** 2
    idx = (torch.arange(n_frames, device=window.device)[:, None] * hop + torch.arange(n_fft, device=window.device)[None, :]).reshape(-1)
    env = torch.zeros((n_frames - 1) * hop + n_fft, dtype=torch.float64, device=window.device)
    env.index_add_(0, idx, w2.repeat(n_frames))
    return env.to(dtype)


_nola_cache = {}


def _nola_ok(window: torch.Tensor, n_fft: int, hop: int) -> bool:
    """torch.istft's NOLA check, cached per window CONTENT (a data_ptr key goes stale when the
    allocator reuses the address for another window)."""
    w = window.detach().to("cpu", torch.float64).contiguous()
    key = (w.numpy().tobytes(), n_fft, hop)
    ok = _nola_cache.get(key)
    if ok is None:
        import numpy as np
        w2 = w.numpy() ** 2
        reps = -(-n_fft // hop)
        env = np.zeros(hop)
        for j in range(reps):
            seg = w2[j * hop: (j + 1) * hop]
            env[: len(seg)] += seg
        ok = _nola_cache[key] = bool(env.min() > 1e-11)
    return ok


_nola_by_tensor = {}


def _nola_ok_cached(window: torch.Tensor, n_fft: int, hop: int) -> bool:
    """Per-tensor memo in front of the content check: valid while the SAME tensor object is alive
    and unmodified (`_version`), so the steady state costs no device-to-host copy."""
    key = (id(window), window._version, n_fft, hop)
    hit = _nola_by_tensor.get(key)
    if hit is not None and hit[0]() is window:
        return hit[1]
    import weakref
    ok = _nola_ok(window, n_fft, hop)
    if len(_nola_by_tensor) > 256:
        _nola_by_tensor.clear()
    _nola_by_tensor[key] = (weakref.ref(window), ok)
    return ok


def istft_fused_supported(n_fft: int, hop: int) -> bool:
    """True when at_istft_f32 takes its fused single-pass path (power-of-two n_fft <= 2048,
    hop = n_fft / {2,4,8,16})."""
    return stft_fused_supported(n_fft) and any(hop * r == n_fft for r in (2, 4, 8, 16))


def istft_tiled_supported(n_fft: int, hop: int) -> bool:
    """True when at_istft_f32 takes the tiled single-pass path of the 96 / 192 kHz default sizes (n_fft 4096 / 8192 with
    hop = n_fft / 4; csrc/stft_generic.hip); like the fused path it takes virtual lead / trail frames without a copy."""
    return n_fft in (4096, 8192) and hop * 4 == n_fft and _knob("AT_ISTFT_TILED_OFF", "0") in ("", "0")


def istft_edit_supported(n_fft: int, hop: int) -> bool:
    """The fused inverse kernel can apply a pending STFT-domain edit while it reads the spectrum."""
    return istft_fused_supported(n_fft, hop) and hop * 4 == n_fft and 64 <= n_fft <= 2048


def istft(stft_bcfn: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, length: int,
          lead: int = 0, trail: int = 0, edit: dict = None) -> torch.Tensor:
    """Inverse STFT of a (B, C, F, N) complex64 HIP tensor -> (B, C, length) float32
    (torch.istft(center=True) semantics).  ``lead`` / ``trail`` all-zero frames are added in front
    of / behind the N frames without copying (match_stride, audio_signal.py:1278-1281).
    ``edit``: a pending STFT-domain edit (``filters.SpecEdit.fused``) applied to the spectrum as it is read
    (``at_istft_edit_f32``); requires :func:`istft_edit_supported`."""
    _require_native_ok(stft_bcfn)
    B, C, F, N = stft_bcfn.shape
    assert F == n_fft // 2 + 1
    if (lead or trail) and not (istft_fused_supported(n_fft, hop) or istft_tiled_supported(n_fft, hop)):
        stft_bcfn = torch.nn.functional.pad(stft_bcfn, (lead, trail))
        N, lead, trail = N + lead + trail, 0, 0
    # physical (B, C, N, F) bin-contiguous layout; a no-op for tensors produced by stft_mel()
    X = stft_bcfn.transpose(2, 3).contiguous()
    dev = X.device
    # torch.istft refuses windows whose overlap-add envelope vanishes (NOLA)
    if not _nola_ok_cached(window, n_fft, hop):
        raise RuntimeError("istft: window overlap add min is (nearly) zero -- the STFT is not invertible")
    lib = _native.lib()
    n_frames = lead + N + trail
    need = int(lib.at_istft_workspace_bytes(B * C, n_frames, n_fft, hop))
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    tw = tables.stft_twiddles(n_fft, dev)
    Xr = torch.view_as_real(X)

    def alloc():
        return torch.empty((B, C, length), dtype=torch.float32, device=dev)

    def launch(out):
        if edit is not None:
            code = lib.at_istft_edit_f32(_native.ptr(Xr), B * C, N, _native.ptr(window), _native.ptr(tw), n_fft, hop,
                                         lead, n_frames, length, _native.ptr(out), _native.ptr(ws), need, int(edit["kind"]), C,
                                         _native.ptr(edit.get("lo")), _native.ptr(edit.get("hi")), _native.ptr(edit.get("shift")),
                                         _native.ptr(edit.get("cut")), _native.ptr(edit.get("maxpow")),
                                         float(edit.get("fill_re", 0.0)), float(edit.get("fill_im", 0.0)),
                                         float(edit.get("top_db", 0.0)), int(edit.get("use_top", 0)), float(edit.get("val", 0.0)),
                                         _native.current_stream(dev))
            _native.check(code, "at_istft_edit_f32")
        else:
            code = lib.at_istft_f32(_native.ptr(Xr), B * C, N, _native.ptr(window), _native.ptr(tw), n_fft, hop,
                                    lead, n_frames, length, _native.ptr(out), _native.ptr(ws), need, _native.current_stream(dev))
            _native.check(code, "at_istft_f32")

    # the inverse follows the placement of its SIGNAL buffer (11 % between allocations at B = 512, tools/placement_survey.py):
    # the same pool as the forward transform's spectrum, fresh-allocation semantics (_PlacedOutputs)
    nbytes = B * C * length * 4
    got = None
    if nbytes >= _PlacedOutputs.MIN_BYTES:
        key = (dev, (B, C, length), torch.cuda.current_stream(dev).cuda_stream, "istft", n_fft, hop, N)
        got = _placed_outputs.acquire(key, nbytes, alloc, launch)
    if got is not None:
        out, launched = got       # (an alias the pool made under its lock: it tells by the storage's use count whether a result is still held)
    else:
        out, launched = _alloc_retry(alloc), False
    if not launched:
        launch(out)
    return out


def stft_adjoint(grad_bcfn: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, T: int) -> torch.Tensor:
    """Backward pass of ``stft_mel(audio (B, C, T), ..., pad=0)``: ``grad_bcfn`` is dL/dX of logical
    shape (B, C, F, N); returns dL/daudio (B, C, T).  Native OLA of the windowed adjoint DFT over the
    centre-padded signal, then the two reflected margins are folded back (torch.stft center=True)."""
    _require_native_ok(grad_bcfn)
    B, C, F, N = grad_bcfn.shape
    assert F == n_fft // 2 + 1 and istft_fused_supported(n_fft, hop)
    G = grad_bcfn.transpose(2, 3).contiguous()   # no copy when the grad has the layout of the forward output
    dev = G.device
    half = n_fft // 2
    Lp = (N - 1) * hop + n_fft
    full = T + n_fft                      # Lp <= full; the tail no frame covers has zero gradient
    out = torch.empty((B, C, full), dtype=torch.float32, device=dev)
    if full > Lp:
        out[..., Lp:] = 0
    tw = tables.stft_twiddles(n_fft, dev)
    code = _native.lib().at_stft_adjoint_f32(_native.ptr(torch.view_as_real(G)), B * C, N, _native.ptr(window),
                                             _native.ptr(tw), n_fft, hop, _native.ptr(out), full,
                                             _native.current_stream(dev))
    _native.check(code, "at_stft_adjoint_f32")
    g = out[..., half: half + T].clone()
    # reflect padding: padded position j < half mirrors x[half - j]; position half + T + j mirrors x[T - 2 - j]
    g[..., 1: half + 1] += out[..., :half].flip(-1)
    g[..., T - 1 - half: T - 1] += out[..., half + T: half + T + half].flip(-1)
    return g


def _fold_pad(out: torch.Tensor, T: int, left: int, right: int, mode: str) -> torch.Tensor:
    """Gradient w.r.t. a signal padded by (left, right) samples in ``mode`` (torch.nn.functional.pad names: the outer padding
    of match_stride, audio_signal.py:1192-1196) -> gradient w.r.t. the signal."""
    g = out[..., left: left + T].clone()
    if (left == 0 and right == 0) or mode == "constant":
        return g
    lo, hi = out[..., :left], out[..., left + T:]
    if mode == "reflect":           # padded position left - 1 - j mirrors x[1 + j]; position left + T + j mirrors x[T - 2 - j]
        if left:
            g[..., 1: left + 1] += lo.flip(-1)
        if right:
            g[..., T - 1 - right: T - 1] += hi.flip(-1)
    elif mode == "replicate":
        if left:
            g[..., 0] += lo.sum(-1)
        if right:
            g[..., T - 1] += hi.sum(-1)
    elif mode == "circular":
        if left:
            g[..., T - left:] += lo
        if right:
            g[..., :right] += hi
    else:
        raise NotImplementedError(f"Unrecognised padding mode {mode}")
    return g


def stft_adjoint_general_supported(window: torch.Tensor, n_fft: int, hop: int) -> bool:
    """Every transform size some forward kernel covers has a native adjoint as long as the inverse kernels take the
    (window, hop) pair (torch.istft's NOLA condition)."""
    return stft_native_supported(n_fft) and 0 < hop <= n_fft and _nola_ok_cached(window, n_fft, hop)


def stft_adjoint_general(grad_bcfn: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, T: int, pad: int = 0,
                         right_pad: int = 0, padding_type: str = "reflect", match_stride: bool = False,
                         _istft=None) -> torch.Tensor:
    """Backward pass of ``stft_mel(audio (B, C, T), ..., pad, right_pad, padding_type, match_stride)`` for EVERY native
    transform size: the run-time sizes (4096 / 8192, 400 / 1200 / 1920 ...) and ``match_stride`` (two frames dropped per
    side, outer padding) included.  The adjoint of one frame's one-sided DFT is N irfft(Y) with Y = G / 2 on the interior
    bins and Re G at DC / Nyquist, the adjoint of framing is window + overlap-add: N x envelope x istft(Y), i.e. the inverse
    kernels with their envelope division undone.  The fused sizes take the dedicated adjoint kernel (at_stft_adjoint_f32:
    no envelope round trip); dropped edge frames are zero frames, the centre reflection and the outer padding are folded
    back onto the samples they copied."""
    B, C, F, n_out = grad_bcfn.shape
    frame_lo, n_expect = stft_frames(T, n_fft, hop, pad, right_pad, match_stride)
    assert F == n_fft // 2 + 1 and n_out == n_expect, (grad_bcfn.shape, n_expect)
    T2 = T + 2 * pad + right_pad
    n_total = 1 + T2 // hop
    drop_hi = n_total - frame_lo - n_out
    half = n_fft // 2
    if _istft is None and istft_fused_supported(n_fft, hop):
        G = grad_bcfn if not (frame_lo or drop_hi) else torch.nn.functional.pad(grad_bcfn, (frame_lo, drop_hi))
        g2 = stft_adjoint(G, window, n_fft, hop, T2)
    else:
        istft_fn = istft if _istft is None else _istft
        Y = grad_bcfn * 0.5
        Y[:, :, 0, :] = grad_bcfn[:, :, 0, :].real.to(Y.dtype)
        Y[:, :, half, :] = grad_bcfn[:, :, half, :].real.to(Y.dtype)
        # K zero frames in front move padded position 0 to output index K hop - half >= 0 of the centre-trimmed inverse;
        # K behind keep the envelope of the last real frames what it is inside the range that is read
        K = -(-half // hop)
        lead, trail = K + frame_lo, drop_hi + K
        n_all = lead + n_out + trail
        full = T2 + n_fft                         # the centre-padded, outer-padded signal
        length = full + K * hop - half
        x = istft_fn(Y, window, n_fft, hop, length, lead=lead, trail=trail)
        env = _ola_envelope(window, hop, n_all, x.dtype)
        env = env[half: half + length]
        if env.numel() < length:
            env = torch.nn.functional.pad(env, (0, length - env.numel()))
        x = torch.where(env > 1e-11, x * (env * float(n_fft)), torch.zeros_like(x))
        g2 = _fold_reflect(x[..., K * hop - half:], T2, half)
    return _fold_pad(g2, T, pad, pad + right_pad, padding_type)


def stft_mel_adjoint_supported(n_fft: int, hop: int, n_mels: int) -> bool:
    return 64 <= n_fft <= 2048 and hop * 4 == n_fft and n_mels <= 8 * (n_fft // 32)


def _fold_reflect(out: torch.Tensor, T: int, half: int) -> torch.Tensor:
    """Gradient w.r.t. the centre-padded signal -> gradient w.r.t. the signal (reflect padding:
    padded position j < half mirrors x[half - j]; position half + T + j mirrors x[T - 2 - j])."""
    g = out[..., half: half + T].clone()
    g[..., 1: half + 1] += out[..., :half].flip(-1)
    g[..., T - 1 - half: T - 1] += out[..., half + T: half + T + half].flip(-1)
    return g


def stft_mel_adjoint(X_bcfn: torch.Tensor, gmel_bcmn: torch.Tensor, bin_table, window: torch.Tensor, n_fft: int,
                     hop: int, T: int) -> torch.Tensor:
    """Backward of the fused mel path given only dL/dmel (B, C, n_mels, N) and the saved spectrum
    (B, C, F, N): dL/daudio (B, C, T) in one kernel (the spectrum gradient is never materialised)."""
    _require_native_ok(X_bcfn)
    B, C, F, N = X_bcfn.shape
    n_mels = gmel_bcmn.shape[2]
    X = X_bcfn.transpose(2, 3).contiguous()
    gm = gmel_bcmn.transpose(2, 3).to(torch.float32).contiguous()
    bands, w = bin_table
    dev = X.device
    half = n_fft // 2
    Lp = (N - 1) * hop + n_fft
    full = T + n_fft
    out = torch.empty((B, C, full), dtype=torch.float32, device=dev)
    if full > Lp:
        out[..., Lp:] = 0
    tw = tables.stft_twiddles(n_fft, dev)
    code = _native.lib().at_stft_mel_adjoint_f32(_native.ptr(torch.view_as_real(X)), _native.ptr(gm), _native.ptr(bands),
                                                 _native.ptr(w), n_mels, B * C, N, _native.ptr(window), _native.ptr(tw),
                                                 n_fft, hop, _native.ptr(out), full, _native.current_stream(dev))
    _native.check(code, "at_stft_mel_adjoint_f32")
    return _fold_reflect(out, T, half)


def istft_adjoint_supported(n_fft: int, hop: int) -> bool:
    return istft_fused_supported(n_fft, hop) and (n_fft // 2) % hop == 0 and stft_fused_supported(n_fft)


def istft_adjoint(grad_bct: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, n_frames: int) -> torch.Tensor:
    """Backward pass of ``istft(X (B, C, F, n_frames)) -> (B, C, T)``: dL/dX of logical shape
    (B, C, F, n_frames).  The adjoint of overlap-add + envelope division + c2r is the FORWARD kernel:
    X_bar[f, k] = (c_k / N) sum_n w[n] u[f hop + n] e^{-2 pi i k n / N},  u = (g / envelope) embedded
    at offset N/2 in zeros, c_k = 2 for interior bins and 1 for DC / Nyquist (what torch's
    fft_c2r backward computes)."""
    _require_native_ok(grad_bct)
    B, C, T = grad_bct.shape
    half = n_fft // 2
    dev = grad_bct.device
    env = _ola_envelope(window, hop, n_frames)
    inv_env = torch.where(env > 1e-11, 1.0 / env, torch.zeros_like(env))
    seg = inv_env[half: half + T]
    if seg.numel() < T:              # samples no frame covers: zero output, zero gradient
        seg = torch.nn.functional.pad(seg, (0, T - seg.numel()))
    v = (grad_bct * (seg * (2.0 / n_fft))).contiguous()
    # enough zeros on the right that (a) the frame count fits and (b) no frame reaches the reflected margin
    right = max(0, (n_frames + 1) * hop - half - T, (n_frames - 1) * hop - T)
    X, _ = stft_mel(v, window, n_fft, hop, pad=half, right_pad=right, padding_type="constant",
                    frame_range=(half // hop, n_frames))
    X[:, :, 0, :] *= 0.5
    X[:, :, half, :] *= 0.5
    return X


def lufs_block_params(rate: int, block_size: float):
    """(K, S) exactly as loudness.py:165-170 computes them (Python floats)."""
    overlap = 0.75
    step = 1.0 - overlap
    K = int(block_size * rate)
    S = int(block_size * rate * step)
    return K, S


_ws_cache = {}


def _workspace(nbytes: int, device):
    """Scratch buffer cached per (device, stream): kernels on different streams never share one
    (a single per-device buffer raced when two streams measured loudness concurrently)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        while len(_ws_cache) > _WS_MAX:          # streams come and go: keep the most recent few
            _ws_cache.pop(next(iter(_ws_cache)))
    return buf


_WS_MAX = 8


def integrated_loudness(audio_bct: torch.Tensor, rate: int, filter_class: str = "K-weighting",
                        block_size: float = 0.400, floor_db: float = float("nan")) -> torch.Tensor:
    """BS.1770 integrated loudness of a (B, C, T) float32 HIP tensor -> (B,) float32."""
    _require_native_ok(audio_bct)
    B, C, T = audio_bct.shape
    if C > 5:
        raise RuntimeError("BS.1770 channel gains are defined for at most 5 channels (loudness.py:49)")
    audio_bct = audio_bct.contiguous()
    dev = audio_bct.device
    sos, gains = tables.weighting_sos(int(rate), filter_class)
    K, S = lufs_block_params(rate, block_size)
    if K <= 0 or S <= 0:
        raise ValueError("block_size * rate must be at least 4 samples")
    warm = tables.lufs_warmup(sos)
    inv_norm = 1.0 / (block_size * rate)
    lib = _native.lib()
    need = lib.at_lufs_workspace_bytes(B, C, T, K, S)
    if need < 0:
        _native.check(int(need), "at_lufs_workspace_bytes")
    ws = _workspace(int(need), dev)
    out = torch.empty((B,), dtype=torch.float32, device=dev)
    code = lib.at_lufs_f32(_native.ptr(audio_bct), B, C, T, sos.ctypes.data, gains.ctypes.data, len(sos), K, S,
                           inv_norm, floor_db, min(warm, 1 << 30), _native.ptr(out), _native.ptr(ws),
                           ws.numel(), _native.current_stream(dev))
    _native.check(code, "at_lufs_f32")
    return out


def integrated_loudness_fir(audio_bct: torch.Tensor, rate: int, firs: np.ndarray, gains, block_size: float = 0.400,
                            floor_db: float = float("nan")) -> torch.Tensor:
    """BS.1770 loudness with the weighting stages as truncated impulse responses -- the reference's FIR approximation
    (loudness.py:69-100: zero padding, ``fft_conv1d`` with the reversed impulse response, gain, ``[1 : nt + 1]``, i.e. the
    causal filter y[n] = g sum_i h[i] x[n - i] with a zero initial state) -- on a (B, C, T) float32 HIP tensor.
    ``firs`` (n_stages, L): impulse responses h, ``gains`` the pass-band gains.  Each stage is one block-FFT launch of
    ``at_fir_fft_f32`` (taps[j] = h[L - 1 - j], centre L - 1: what the reference convolves with); its replicate padding
    differs from zero padding in the first L - 1 outputs by x[0] * sum_{i > n} h[i], taken off on that slice.  Hop
    energies and gating: ``at_lufs_f32`` with a pass-through stage."""
    _require_native_ok(audio_bct)
    B, C, T = audio_bct.shape
    x = audio_bct.contiguous()
    dev = x.device
    tw = tables.stft_twiddles(2048, dev)
    lib = _native.lib()
    for h, g in zip(np.asarray(firs, dtype=np.float64), gains):
        L = int(h.shape[0])
        Lp = _pad8(L)
        half = L - 1
        tp_np = np.zeros((1, Lp), dtype=np.float32)
        tp_np[0, :L] = (float(g) * h[::-1]).astype(np.float32)
        tail = np.concatenate([np.cumsum(h[::-1])[::-1][1:], [0.0]]) * float(g)       # sum_{i > n} h[i], n = 0 .. L - 1
        key = ("lufs_fir", int(rate), L, hash(tp_np.tobytes()))
        tp, fix = tables.device_table(key, dev, lambda: (tp_np, tail[: L - 1].astype(np.float32)))
        y = torch.empty_like(x)
        code = lib.at_fir_fft_f32(_native.ptr(x), B, C, T, _native.ptr(tp), 1, Lp, half, 0, _native.ptr(tw), _native.ptr(y),
                                  _native.current_stream(dev))
        _native.check(code, "at_fir_fft_f32")
        n = min(L - 1, T)
        y[..., :n] -= x[..., :1] * fix[:n]
        x = y
    ident = np.asarray([[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]], dtype=np.float64)
    one = np.asarray([1.0], dtype=np.float64)
    K, S = lufs_block_params(rate, block_size)
    if K <= 0 or S <= 0:
        raise ValueError("block_size * rate must be at least 4 samples")
    need = lib.at_lufs_workspace_bytes(B, C, T, K, S)
    if need < 0:
        _native.check(int(need), "at_lufs_workspace_bytes")
    ws = _workspace(int(need), dev)
    out = torch.empty((B,), dtype=torch.float32, device=dev)
    code = lib.at_lufs_f32(_native.ptr(x), B, C, T, ident.ctypes.data, one.ctypes.data, 1, K, S, 1.0 / (block_size * rate),
                           floor_db, 0, _native.ptr(out), _native.ptr(ws), ws.numel(), _native.current_stream(dev))
    _native.check(code, "at_lufs_f32")
    return out


def have(symbol: str) -> bool:
    """Diagnostic only (never a dispatch switch): the loader already refuses a library that lacks
    any declared entry point, so there is no silent fallback to route around a missing kernel."""
    return hasattr(_native.lib(), symbol)


# ------------------------------------------------------------------ FIR family
def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


FIR_FFT_MIN_TAPS = 96   # below this the direct register-window kernel is cheaper than two block FFTs
# Above this the overlap-save kernel's partitions (one pass over the signal per 1024 taps: a 50 Hz high-pass at 44.1 kHz
# has 44 983 taps = 44 launches, 17 ms for 256 x 5 s) cost more than ONE circular convolution by the four-step FFT at a
# length >= T + L - 1 (four launches + the padding copy, ~6 passes): HighPass's default cutoffs 50 / 100 / 250 Hz.
FIR_LONG_MIN_TAPS = 6145


@functools.lru_cache(256)
def _fir_long_length(n: int):
    """Smallest even length >= n the four-step convolution has a plan for (None beyond its range)."""
    lib = _native.lib()
    m = n + (n & 1)
    for _ in range(8192):
        if m > (1 << 21):
            return None
        if lib.at_longconv_supported(m):
            return m
        m += 2
    return None


def _fir_long(audio, tp, rows, L, highpass, out):
    """Per-item FIR of thousands of taps as one circular convolution (``at_longconv_room_f32``): the signal replicate-padded
    by half a filter on either side and zero-filled to a planned length N >= T + L - 1, the flipped taps as the impulse
    response read rotated left by L - 1 -- output sample n of the filter is then sample n of the convolution (no wrap
    reaches the first T samples).  Same values as the direct / overlap-save kernels up to float32 summation order."""
    B, C, T = audio.shape
    H = (L - 1) // 2
    N = _fir_long_length(T + L - 1)
    xp = torch.empty((B, C, N), dtype=torch.float32, device=audio.device)
    xp[..., H: H + T] = audio
    xp[..., :H] = audio[..., :1]
    xp[..., H + T: T + 2 * H] = audio[..., -1:]
    xp[..., T + 2 * H:] = 0
    ir = tp[:, :L].flip(-1).reshape(rows, 1, L)
    if rows != B:
        ir = ir.expand(B, 1, L)
    shift = torch.full((B, 1), L - 1, dtype=torch.int64, device=audio.device)
    y = room_convolve(xp, ir, shift, None)[..., :T]
    if out is None:
        return (audio - y) if highpass else y.contiguous()
    if highpass:
        torch.sub(audio, y, out=out)
    else:
        out.copy_(y)
    return out


def fir_per_item(audio: torch.Tensor, taps: torch.Tensor, highpass: bool = False, replicate: bool = True,
                 method: str = "auto", out: torch.Tensor = None, L: int = None):
    """Per-item FIR with replicate padding.  ``taps`` (B or 1, L) odd-length, centred.
    ``method``: "direct" (at_fir_per_item_f32), "fft" (overlap-save, at_fir_fft_f32) or "auto".
    ``out``: optional contiguous float32 result buffer of ``audio``'s shape (e.g. one slab of a band stack).
    ``L``: ``taps`` is already the zero-padded (rows, pad8(L)) float32 device table of filters of length L
    (:func:`sinc_taps_native` / :func:`eq_taps_native`): no padding copy."""
    _require_native_ok(audio)
    assert replicate
    B, C, T = audio.shape
    audio = audio.contiguous()
    if L is None:
        rows, L = taps.shape
        assert L % 2 == 1 and rows in (1, B)
        Lp = _pad8(L)
        tp = torch.zeros((rows, Lp), dtype=torch.float32, device=audio.device)
        tp[:, :L] = taps.to(audio.device, torch.float32)
    else:
        rows, Lp = taps.shape
        assert L % 2 == 1 and rows in (1, B) and Lp == _pad8(L) and taps.is_contiguous() and taps.dtype == torch.float32
        tp = taps
    half = (L - 1) // 2
    if out is None:
        out = torch.empty_like(audio)
    assert out.shape == audio.shape and out.is_contiguous() and out.dtype == torch.float32 and out.device == audio.device
    if method == "auto":
        method = _knob("AT_FIR_METHOD", None)               # development A/B (dev builds only): direct | fft | long
        if not method:
            method = "fft" if L >= FIR_FFT_MIN_TAPS else "direct"
            if L >= FIR_LONG_MIN_TAPS and _LONGCONV and _fir_long_length(T + L - 1) is not None:
                method = "long"
    if method == "long" and _fir_long_length(T + L - 1) is None:
        method = "fft"                                       # (forced, but no planned length: the overlap-save form)
    if method == "long":
        return _fir_long(audio, tp, rows, L, highpass, out)
    if method == "fft":
        tw = tables.stft_twiddles(2048, audio.device)
        code = _native.lib().at_fir_fft_f32(_native.ptr(audio), B, C, T, _native.ptr(tp), rows, Lp, half,
                                            1 if highpass else 0, _native.ptr(tw), _native.ptr(out),
                                            _native.current_stream(audio.device))
        _native.check(code, "at_fir_fft_f32")
        return out
    code = _native.lib().at_fir_per_item_f32(_native.ptr(audio), B, C, T, _native.ptr(tp), rows, Lp, half,
                                             1 if highpass else 0, _native.ptr(out), _native.current_stream(audio.device))
    _native.check(code, "at_fir_per_item_f32")
    return out


# ------------------------------------------------------------ stft_data edits
def spec_native(X: torch.Tensor) -> bool:
    """stft_data the in-place edit kernels accept: complex64 HIP tensor of logical shape
    (B, C, F, N) in the bin-contiguous physical layout stft_mel() produces, no autograd."""
    return (X is not None and X.is_cuda and X.dtype == torch.complex64 and X.ndim == 4
            and not (X.requires_grad and torch.is_grad_enabled()) and X.transpose(2, 3).is_contiguous())


def _spec_out(X: torch.Tensor):
    """(fresh result tensor of the same physical layout, its real view, the input's real view, sizes):
    the reference's edits return NEW tensors (the old stft_data a caller may still hold must not
    change), so the kernels read the input and write the result in one pass."""
    B, C, F, N = X.shape
    Y = torch.empty((B, C, N, F), dtype=X.dtype, device=X.device)
    return Y.transpose(2, 3), torch.view_as_real(Y), torch.view_as_real(X.transpose(2, 3)), B, C, N, F


def spec_mask(X: torch.Tensor, axis: int, lo: torch.Tensor, hi: torch.Tensor, grid: torch.Tensor, val: float):
    Y, Yr, Xr, B, C, N, F = _spec_out(X)
    lo = lo.reshape(-1).to(X.device, torch.float64).expand(B).contiguous()
    hi = hi.reshape(-1).to(X.device, torch.float64).expand(B).contiguous()
    grid = grid.to(X.device, torch.float32).contiguous()
    assert grid.numel() == (F if axis == 0 else N)
    v = float(val)
    # magnitude = phase = val, as float32 ops: val * exp(1j * val)
    fill = torch.tensor(v, dtype=torch.float32) * torch.exp(1j * torch.tensor(v, dtype=torch.float32))
    code = _native.lib().at_spec_mask_f32(_native.ptr(Xr), _native.ptr(Yr), B, C, N, F, axis, _native.ptr(lo), _native.ptr(hi),
                                          _native.ptr(grid), float(fill.real), float(fill.imag),
                                          _native.current_stream(X.device))
    _native.check(code, "at_spec_mask_f32")
    return Y


def spec_mask_fill(val: float):
    """(re, im) of ``val * exp(1j * val)`` as float32 ops: magnitude = phase = val (dsp.py:252-258)."""
    v = float(val)
    fill = torch.tensor(v, dtype=torch.float32) * torch.exp(1j * torch.tensor(v, dtype=torch.float32))
    return float(fill.real), float(fill.imag)


def mask_edit(kind: int, lo: torch.Tensor, hi: torch.Tensor, grid: torch.Tensor, val: float, B: int, device) -> dict:
    """Arguments of ``at_istft_edit_f32`` for a frequency (kind 1) / time (kind 2) mask: the per-item index ranges
    ``lo <= grid[i] < hi`` selects.  The grid is monotone, so the selected set is the contiguous range between the
    first index with ``grid >= lo`` and the first with ``grid >= hi`` -- evaluated in float64 like spec_mask_kernel."""
    g = grid.to(device, torch.float64).contiguous()
    lo64 = lo.reshape(-1).to(device, torch.float64).expand(B).contiguous()
    hi64 = hi.reshape(-1).to(device, torch.float64).expand(B).contiguous()
    re, im = spec_mask_fill(val)
    return {"kind": kind, "lo": torch.searchsorted(g, lo64).to(torch.int32), "hi": torch.searchsorted(g, hi64).to(torch.int32),
            "fill_re": re, "fill_im": im}


def spec_maxpow(X: torch.Tensor) -> torch.Tensor:
    """max |X|^2 over the whole (batch) spectrum as a 1-element float32 device tensor (log_magnitude's global
    top_db floor, audio_signal.py:1486)."""
    B, C, F, N = X.shape
    Xr = torch.view_as_real(X.transpose(2, 3))
    mp = torch.empty(1, dtype=torch.float32, device=X.device)
    _native.check(_native.lib().at_spec_maxpow_f32(_native.ptr(Xr), B * C * N * F, _native.ptr(mp), _native.current_stream(X.device)),
                  "at_spec_maxpow_f32")
    return mp


def spec_gate(X: torch.Tensor, thr_db: torch.Tensor, amount: torch.Tensor, tf: torch.Tensor, tt: torch.Tensor):
    """``X * (1 - amount * conv2d(20 log10(max(|X|, 1e-4)) < thr_db, outer(tf, tt)))`` for a native (B, C, F, N)
    spectrum (``at_spec_gate_f32``).  ``thr_db`` (B or 1, C, F); ``amount`` broadcastable to (B,); ``tf`` / ``tt`` the
    two 1-D factors of the normalised smoothing filter."""
    Y, Yr, Xr, B, C, N, F = _spec_out(X)
    thr = thr_db.to(X.device, torch.float32).reshape(-1, C, F).contiguous()
    assert thr.shape[0] in (1, B)
    amt = amount.to(X.device, torch.float32).reshape(-1).expand(B).contiguous()
    tf = tf.to(X.device, torch.float32).contiguous()
    tt = tt.to(X.device, torch.float32).contiguous()
    code = _native.lib().at_spec_gate_f32(_native.ptr(Xr), _native.ptr(Yr), B, C, N, F, _native.ptr(thr), 1 if thr.shape[0] == B and B > 1 else 0,
                                          _native.ptr(amt), _native.ptr(tf), int(tf.numel()), _native.ptr(tt), int(tt.numel()),
                                          _native.current_stream(X.device))
    _native.check(code, "at_spec_gate_f32")
    return Y


def spec_phase_shift(X: torch.Tensor, shift: torch.Tensor):
    Y, Yr, Xr, B, C, N, F = _spec_out(X)
    sh = shift.reshape(-1).to(X.device, torch.float32).expand(B).contiguous()
    code = _native.lib().at_spec_phase_shift_f32(_native.ptr(Xr), _native.ptr(Yr), B, C, N, F, _native.ptr(sh),
                                                 _native.current_stream(X.device))
    _native.check(code, "at_spec_phase_shift_f32")
    return Y


def spec_polar_elem(X: torch.Tensor, b: torch.Tensor, a: torch.Tensor = None):
    """Per-element polar edit of a native spectrum: ``X * exp(1j * b)`` (``a`` None) or ``|a| * exp(1j * b)``
    where ``X == 0`` and ``X`` elsewhere (|a|: the reference's phase setter re-derives the magnitude).  ``a`` / ``b``: float32, broadcastable to X's logical shape."""
    Y, Yr, Xr, B, C, N, F = _spec_out(X)
    shape = (B, C, F, N)
    bb = b.to(X.device, torch.float32).expand(shape).contiguous()
    aa = None if a is None else a.to(X.device, torch.float32).expand(shape).contiguous()
    code = _native.lib().at_spec_polar_elem_f32(_native.ptr(Xr), _native.ptr(Yr), B, C, N, F, _native.ptr(aa), _native.ptr(bb),
                                                0 if a is None else 1, _native.current_stream(X.device))
    _native.check(code, "at_spec_polar_elem_f32")
    return Y


def spec_mask_lowmag(X: torch.Tensor, cutoff_db: torch.Tensor, val: float, top_db=80.0):
    Y, Yr, Xr, B, C, N, F = _spec_out(X)
    cut = cutoff_db.reshape(-1).to(X.device, torch.float64).expand(B).contiguous()
    mp = torch.empty(1, dtype=torch.float32, device=X.device)
    lib = _native.lib()
    st = _native.current_stream(X.device)
    _native.check(lib.at_spec_maxpow_f32(_native.ptr(Xr), B * C * N * F, _native.ptr(mp), st), "at_spec_maxpow_f32")
    code = lib.at_spec_mask_lowmag_f32(_native.ptr(Xr), _native.ptr(Yr), B, C, N, F, _native.ptr(cut), _native.ptr(mp),
                                       float(top_db if top_db is not None else 0.0), 0 if top_db is None else 1,
                                       float(val), st)
    _native.check(code, "at_spec_mask_lowmag_f32")
    return Y


def absmax(x: torch.Tensor, want_index: bool = False):
    """Per-row peak of a (..., T) HIP tensor in one pass: max |x| over the last axis (and the
    first index attaining it) -- x.abs().max(-1) / x.abs().argmax(-1) of effects.py:100,118,160."""
    _require_native_ok(x)
    x = x.contiguous()
    T = x.shape[-1]
    rows = x.numel() // T
    vmax = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    imax = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device) if want_index else None
    code = _native.lib().at_absmax_f32(_native.ptr(x), rows, T, _native.ptr(vmax), _native.ptr(imax),
                                       _native.current_stream(x.device))
    _native.check(code, "at_absmax_f32")
    return (vmax, imax) if want_index else vmax


def roll_pad(x: torch.Tensor, shift: torch.Tensor, T: int) -> torch.Tensor:
    """(B, C, L) -> (B, C, T): rows zero-padded / truncated to T and rotated left by ``shift`` (B, C)."""
    _require_native_ok(x)
    x = x.contiguous()
    B, C, L = x.shape
    out = torch.empty((B, C, T), dtype=torch.float32, device=x.device)
    sh = None if shift is None else shift.to(torch.int64).contiguous()
    code = _native.lib().at_roll_pad_f32(_native.ptr(x), B * C, L, _native.ptr(sh), T, _native.ptr(out),
                                         _native.current_stream(x.device))
    _native.check(code, "at_roll_pad_f32")
    return out


def quantize(x: torch.Tensor, q: torch.Tensor, mulaw: bool) -> torch.Tensor:
    """``quantization`` / ``mulaw_quantization`` of a (B, C, T) HIP tensor in one pass (``at_quantize_f32``); ``q`` (B,)
    float32: the channel count resp. mu = channels - 1 of every item."""
    _require_native_ok(x)
    x = x.contiguous()
    B = x.shape[0]
    q = q.reshape(-1).to(x.device, torch.float32).expand(B).contiguous()
    out = torch.empty_like(x)
    code = _native.lib().at_quantize_f32(_native.ptr(x), B, x.numel() // B, _native.ptr(q), 1 if mulaw else 0, _native.ptr(out),
                                         _native.current_stream(x.device))
    _native.check(code, "at_quantize_f32")
    return out


def collect_windows(x: torch.Tensor, win: int, hop: int) -> torch.Tensor:
    """(rows, T) -> (rows * nw, 1, win): windows of ``win`` samples every ``hop`` along the batch axis (dsp.py:70-108)."""
    _require_native_ok(x)
    x = x.contiguous()
    rows, T = x.shape
    nw = max((T - win) // hop + 1, 0) if T >= win else 0
    out = torch.empty((rows * nw, 1, win), dtype=torch.float32, device=x.device)
    code = _native.lib().at_collect_windows_f32(_native.ptr(x), rows, T, int(win), int(hop), _native.ptr(out),
                                                _native.current_stream(x.device))
    _native.check(code, "at_collect_windows_f32")
    return out


def overlap_add(frames: torch.Tensor, rows: int, hop: int, padded_len: int, trim: int) -> torch.Tensor:
    """(rows * nw, 1, win) windows -> (rows, padded_len - 2 trim): overlap-add divided by the overlap count, trimmed
    (dsp.py:110-151: fold, fold of ones, division, trim -- one gather here)."""
    _require_native_ok(frames)
    frames = frames.contiguous()
    win = frames.shape[-1]
    nw = frames.shape[0] // rows
    out_len = max(int(padded_len) - 2 * int(trim), 0)
    out = torch.empty((rows, out_len), dtype=torch.float32, device=frames.device)
    code = _native.lib().at_overlap_add_f32(_native.ptr(frames), rows, nw, int(win), int(hop), int(trim), out_len,
                                            _native.ptr(out), _native.current_stream(frames.device))
    _native.check(code, "at_overlap_add_f32")
    return out


def alter_drr(x: torch.Tensor, t0: int, drr: torch.Tensor, want_peak: bool = False):
    """alter_drr + ensure_max_of_audio of a batch of impulse responses (B, C, T); ``drr`` (B,).  ``want_peak``: also
    returns what ``absmax(out, want_index=True)`` would (max |out| and its first position per row), found in the kernel's
    output pass."""
    _require_native_ok(x)
    x = x.contiguous()
    B, C, T = x.shape
    d = drr.reshape(-1).to(x.device, torch.float32).contiguous()
    assert d.numel() == B
    out = torch.empty_like(x)
    vmax = torch.empty((B, C), dtype=torch.float32, device=x.device) if want_peak else None
    imax = torch.empty((B, C), dtype=torch.int64, device=x.device) if want_peak else None
    code = _native.lib().at_alter_drr_peak_f32(_native.ptr(x), B, C, T, int(t0), _native.ptr(d), _native.ptr(out),
                                               _native.ptr(vmax), _native.ptr(imax), _native.current_stream(x.device))
    _native.check(code, "at_alter_drr_peak_f32")
    return (out, vmax, imax) if want_peak else out


def sinc_taps_batched(cutoffs: torch.Tensor, zeros: float, host_cutoffs: torch.Tensor = None):
    """Vectorised design of the per-item windowed-sinc low-pass taps of ``low_pass`` /
    ``high_pass`` (dsp.py:177-179 -> julius.LowPassFilter): every item has its own length
    2*half_i+1, ``half_i = int(zeros / c_i / 2)`` in float32 exactly as upstream; the rows are
    centred in a common (B, 2*Hmax+1) array.  Runs on ``cutoffs.device``.

    The range checks (julius raises ValueError) and the common length need the cutoffs on the
    HOST.  ``host_cutoffs`` (the CPU twin that ``util.prepare_batch`` keeps of every small parameter
    tensor) answers them without a device-to-host synchronisation; without it there is one."""
    c = cutoffs.reshape(-1).to(torch.float32)
    pos = c > 0
    half = torch.where(pos, (zeros / torch.where(pos, c, torch.ones_like(c)) / 2).to(torch.int64), torch.zeros_like(c, dtype=torch.int64))
    if host_cutoffs is not None:
        # the same float32 arithmetic on the host (IEEE division on both sides: identical half sizes)
        ch = host_cutoffs.reshape(-1).to(torch.float32)
        posh = ch > 0
        halfh = torch.where(posh, (zeros / torch.where(posh, ch, torch.ones_like(ch)) / 2).to(torch.int64),
                            torch.zeros_like(ch, dtype=torch.int64))
        cmin, cmax, hmax = float(ch.min()), float(ch.max()), float(halfh.max())
    else:
        # one host round trip for the two range checks and the common length
        cmin, cmax, hmax = torch.stack([c.min().double(), c.max().double(), half.max().double()]).tolist()
    if cmin < 0:
        raise ValueError("Minimum cutoff must be larger than zero.")
    if cmax > 0.5:
        raise ValueError("A cutoff above 0.5 does not make sense.")
    H = int(hmax)
    n = torch.arange(-H, H + 1, device=c.device)
    nf = n.to(torch.float32)[None, :]
    hf = half.to(torch.float32)[:, None]
    inside = n[None, :].abs() <= half[:, None]
    # symmetric (non-periodic) Hann of length 2h+1 evaluated on the common grid
    win = 0.5 - 0.5 * torch.cos(2 * math.pi * (nf + hf) / torch.clamp(2 * hf, min=1.0))
    win = torch.where(half[:, None] == 0, torch.ones_like(win), win)
    arg = 2 * c[:, None] * math.pi * nf
    sinc = torch.where(arg == 0, torch.ones_like(arg), torch.sin(arg) / arg)
    h = torch.where(inside, 2 * c[:, None] * win * sinc, torch.zeros_like(arg))
    h = h / h.sum(-1, keepdim=True)
    h = torch.where(pos[:, None], h, torch.zeros_like(h))
    return h


def _sinc_limits(c: torch.Tensor, zeros: float, host_cutoffs: torch.Tensor = None) -> int:
    """Range checks of the cutoffs (julius raises ValueError) and the common half size max_b int(zeros / c_b / 2), from
    the host twin of the cutoffs when there is one (no device-to-host synchronisation), else with one round trip."""
    def halves(v):
        pos = v > 0
        return torch.where(pos, (zeros / torch.where(pos, v, torch.ones_like(v)) / 2).to(torch.int64),
                           torch.zeros_like(v, dtype=torch.int64))
    if host_cutoffs is not None:
        # the same float32 arithmetic as on the device (IEEE division on both sides), in numpy: a handful of torch CPU
        # ops on a 1024-element tensor cost more than the kernel they size on a 256-core host (thread-pool wake-ups)
        import numpy as np
        ch = host_cutoffs.detach().reshape(-1).to(torch.float32).numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            hv = np.where(ch > 0, (np.float32(zeros) / np.where(ch > 0, ch, np.float32(1)) / np.float32(2)).astype(np.int64), 0)
        cmin, cmax, hmax = float(ch.min()), float(ch.max()), float(hv.max())
    else:
        cmin, cmax, hmax = torch.stack([c.min().double(), c.max().double(), halves(c).max().double()]).tolist()
    if cmin < 0:
        raise ValueError("Minimum cutoff must be larger than zero.")
    if cmax > 0.5:
        raise ValueError("A cutoff above 0.5 does not make sense.")
    return int(hmax)


def sinc_taps_native(cutoffs: torch.Tensor, zeros: float, host_cutoffs: torch.Tensor = None):
    """``sinc_taps_batched`` as ONE launch (``at_sinc_taps_f32``): returns ``(table (B, pad8(L)) float32, L)`` -- the
    zero-padded tap table the FIR kernels read, row b centred at column (L - 1) / 2."""
    c = cutoffs.reshape(-1).to(torch.float32).contiguous()
    _require_native_ok(c)
    H = _sinc_limits(c, zeros, host_cutoffs)
    L = 2 * H + 1
    Lp = _pad8(L)
    tp = torch.empty((c.shape[0], Lp), dtype=torch.float32, device=c.device)
    code = _native.lib().at_sinc_taps_f32(_native.ptr(c), c.shape[0], float(zeros), H, Lp, _native.ptr(tp),
                                          _native.current_stream(c.device))
    _native.check(code, "at_sinc_taps_f32")
    return tp, L


def eq_taps_native(weights: torch.Tensor, bank: torch.Tensor, half: int):
    """Composite equalizer FIR per item as ONE launch (``at_eq_taps_f32``): weights (B, n_bands) linear gains, bank
    (n_bands - 1, L) device copy of the band-split low-pass bank.  Returns ``(table (B, pad8(L)), L)``."""
    w = weights.to(torch.float32).contiguous()
    _require_native_ok(w)
    B, n_bands = w.shape
    L = int(bank.shape[1])
    Lp = _pad8(L)
    tp = torch.empty((B, Lp), dtype=torch.float32, device=w.device)
    code = _native.lib().at_eq_taps_f32(_native.ptr(w), _native.ptr(bank), B, n_bands, L, int(half), Lp, _native.ptr(tp),
                                        _native.current_stream(w.device))
    _native.check(code, "at_eq_taps_f32")
    return tp, L


def sinc_filter(audio: torch.Tensor, cutoffs_norm: torch.Tensor, zeros: float, highpass: bool, host_cutoffs=None):
    B = audio.shape[0]
    tp, L = sinc_taps_native(cutoffs_norm.to(audio.device).reshape(B), zeros,
                             None if host_cutoffs is None else host_cutoffs.reshape(B))
    return fir_per_item(audio, tp, highpass=highpass, L=L)


RESAMPLE_LDS_LIMIT = 160 * 1024


def resample_supported(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945) -> bool:
    """True when at_resample_f32 has a tile for this ratio: one thread tile of 4 frames needs
    (4 old + 2 width + LG) floats of LDS with ``old`` gcd-reduced (csrc/fir.hip); reduced rates
    above ~9 k (44100 -> 16001) do not fit 160 KB and take the torch formulation, as every ratio
    julius handles must (audio_signal.py:732)."""
    g = math.gcd(int(old_sr), int(new_sr))
    old, new = int(old_sr) // g, int(new_sr) // g
    if old == new:
        return True
    sr = min(new, old) * rolloff
    width = math.ceil(zeros * old / sr)
    # LG <= taps of the dense bank = 2 width + old
    need = (4 * old + 2 * width + (2 * width + old) + 8) * 4
    return need <= RESAMPLE_LDS_LIMIT


_RESAMPLE_MFMA = _knob("AT_RESAMPLE_MFMA", "1") != "0"     # development A/B switch, read once
_RESAMPLE_F16 = _knob("AT_RESAMPLE_F16", "1") != "0"       # the fp16-split matrix-core form (round 4)


def resample_first_form(old: int, new: int) -> str:
    """Which kernel family ``resample`` tries FIRST for the reduced ratio old -> new ("f16", "mfma" or "grouped"): the one
    predicate the dispatcher below and the table broadcast of dist.broadcast_cfg5_tables share."""
    lib = _native.lib()
    if _RESAMPLE_F16 and lib.at_resample_f16s_supported(old, new):
        return "f16"
    if _RESAMPLE_MFMA and lib.at_resample_mfma_supported(old, new):
        return "mfma"
    return "grouped"


def resample(audio: torch.Tensor, old_sr: int, new_sr: int):
    _require_native_ok(audio)
    g = math.gcd(int(old_sr), int(new_sr))
    old, new = int(old_sr) // g, int(new_sr) // g
    if old == new:
        return audio
    dev = audio.device
    B, C, T = audio.shape
    audio = audio.contiguous()
    out_len = int(math.floor(new * T / old))
    out = torch.empty((B, C, out_len), dtype=torch.float32, device=dev)
    lib = _native.lib()
    form = resample_first_form(old, new)
    if form == "f16" and T >= 16:
        W_np, lo_np, old, new, width, NPB, NC, wk = tables.resample_f16_bank(old, new)
        W, lo = tables.device_table(("resample_f16", old, new), dev, lambda: (W_np.view(np.int32), lo_np))
        code = lib.at_resample_f16s_f32(_native.ptr(audio), B * C, T, _native.ptr(W), _native.ptr(lo), old, new, width,
                                        NPB, NC, int(lo_np.max()), wk, _native.ptr(out), out_len, _native.current_stream(dev))
        if code != -2:            # AT_ERR_UNSUPPORTED (a tile that does not fit): the f32 kernels below
            _native.check(code, "at_resample_f16s_f32")
            return out
    if form != "grouped" and _RESAMPLE_MFMA and lib.at_resample_mfma_supported(old, new):
        W_np, lo_np, old, new, width, NPB, NC = tables.resample_mfma_bank(old, new)
        need = (32 * old + int(lo_np.max()) + 32 * NC + 32 + 4) * 4       # one 32-frame tile of LDS
        if need <= RESAMPLE_LDS_LIMIT:
            W, lo = tables.device_table(("resample_mfma", old, new), dev, lambda: (W_np, lo_np))
            code = lib.at_resample_mfma_f32(_native.ptr(audio), B * C, T, _native.ptr(W), _native.ptr(lo), old, new, width,
                                            NPB, NC, int(lo_np.max()), _native.ptr(out), out_len, _native.current_stream(dev))
            _native.check(code, "at_resample_mfma_f32")
            return out
    wg_np, base_np, old, new, width, NG, LG = tables.resample_grouped_bank(old, new)
    wg, base = tables.device_table(("resample_grouped", old, new), dev, lambda: (wg_np, base_np))
    code = lib.at_resample_f32(_native.ptr(audio), B * C, T, _native.ptr(wg), _native.ptr(base), old, new, width,
                               NG, LG, _native.ptr(out), out_len, _native.current_stream(dev))
    _native.check(code, "at_resample_f32")
    return out


def resample_adjoint(gy: torch.Tensor, old_sr: int, new_sr: int, T: int) -> torch.Tensor:
    """dL/dx (B, C, T) of ``resample`` from dL/dy (B, C, floor(new T / old)): the transposed polyphase sum on the SAME
    kernel (``at_resample_f32`` with the rates swapped and ``tables.resample_adjoint_bank``), then the replicate padding
    folded back (everything the padded positions received goes to x[0] / x[T - 1])."""
    _require_native_ok(gy)
    g = math.gcd(int(old_sr), int(new_sr))
    if int(old_sr) // g == int(new_sr) // g:
        return gy
    wg_np, base_np, old, new, width, NG, LG, J = tables.resample_adjoint_bank(int(old_sr) // g, int(new_sr) // g)
    dev = gy.device
    B, C, n = gy.shape
    # one zero sample on either side: the kernel's replicate padding then repeats zeros
    gin = torch.zeros((B, C, n + 2), dtype=torch.float32, device=dev)
    gin[..., 1: n + 1] = gy
    Lp = T + 2 * width + old
    out = torch.empty((B, C, Lp), dtype=torch.float32, device=dev)
    wg, base = tables.device_table(("resample_adjoint", old, new), dev, lambda: (wg_np, base_np))
    code = _native.lib().at_resample_f32(_native.ptr(gin), B * C, n + 2, _native.ptr(wg), _native.ptr(base), new, old,
                                         J * new - 1, NG, LG, _native.ptr(out), Lp, _native.current_stream(dev))
    _native.check(code, "at_resample_f32 (adjoint)")
    gx = out[..., width: width + T].clone()
    gx[..., 0] += out[..., :width].sum(-1)
    gx[..., T - 1] += out[..., width + T:].sum(-1)
    return gx


def resample_adjoint_supported(old_sr: int, new_sr: int) -> bool:
    """The transposed bank fits the kernel's LDS tile (roles swapped: the input advances by ``new`` per frame)."""
    g = math.gcd(int(old_sr), int(new_sr))
    old, new = int(old_sr) // g, int(new_sr) // g
    if old == new:
        return True
    if not resample_supported(old_sr, new_sr) or old * new > (1 << 22):
        return False
    plan = tables.resample_adjoint_bank(old, new)
    J, LG = plan[7], plan[6]
    return (4 * new + 2 * (J * new - 1) + LG + 8) * 4 <= RESAMPLE_LDS_LIMIT


_LONGCONV = _knob("AT_LONGCONV", "1") != "0"       # development A/B switch, read once


def longconv_supported(T: int) -> bool:
    """True when the hand-written four-step FFT has a plan for the length (``at_longconv_supported``)."""
    return bool(_native.lib().at_longconv_supported(int(T)))


def longconv_enabled(T: int) -> bool:
    return _LONGCONV and longconv_supported(T)


def fftconv(x: torch.Tensor, ir: torch.Tensor, scale: torch.Tensor = None, engine: str = None):
    """Circular convolution of x (B,C,T) with ir (B,1|C,T) at length T, times scale (B,1|C,1).
    ``engine``: "fourstep" (csrc/longconv.hip), "rocfft" (csrc/fftconv.hip) or None = four-step
    when the length has a plan (``AT_LONGCONV=0`` forces rocFFT for A/B runs)."""
    _require_native_ok(x)
    B, C, T = x.shape
    Cir = ir.shape[1]
    x = x.contiguous()
    ir = ir.contiguous()
    if scale is not None:
        scale = scale.reshape(B, Cir).to(torch.float32).contiguous()
    lib = _native.lib()
    if engine is None:
        engine = "fourstep" if longconv_enabled(T) else "rocfft"
    if engine == "fourstep" and (x.data_ptr() % 8 or ir.data_ptr() % 8):
        # the four-step kernels read float2: a rows == 1 view with an odd sample offset stays 4-byte aligned
        # through .contiguous() -- realign instead of failing with AT_ERR_INVALID
        x = x.clone() if x.data_ptr() % 8 else x
        ir = ir.clone() if ir.data_ptr() % 8 else ir
    if engine == "fourstep":
        tb = tables.longconv_tables(T, x.device)
        need = int(lib.at_longconv_workspace_bytes(B, C, Cir, T))
        if need < 0:
            _native.check(need, "at_longconv_workspace_bytes")
        ws = _workspace(need, x.device)
        out = torch.empty_like(x)
        code = lib.at_longconv_circ_f32(_native.ptr(x), _native.ptr(ir), _native.ptr(scale), B, C, Cir, T, _native.ptr(tb),
                                        _native.ptr(out), _native.ptr(ws), ws.numel(), _native.current_stream(x.device))
        _native.check(code, "at_longconv_circ_f32")
        return out
    need = int(lib.at_fftconv_workspace_bytes(B, C, Cir, T))
    if need < 0:
        _native.check(need, "at_fftconv_workspace_bytes")
    ws = _workspace(need, x.device)      # spectra + rocFFT's work buffer, cached per (device, stream)
    out = torch.empty_like(x)
    code = lib.at_fftconv_circ_f32(_native.ptr(x), _native.ptr(ir), _native.ptr(scale), B, C, Cir, T, _native.ptr(out),
                                   _native.ptr(ws), ws.numel(), _native.current_stream(x.device))
    _native.check(code, "at_fftconv_circ_f32")
    return out


def room_convolve(x: torch.Tensor, ir: torch.Tensor, shift: torch.Tensor = None, scale: torch.Tensor = None,
                  want_peaks: bool = False):
    """``at_longconv_room_f32``: circular convolution of x (B,C,T) with the impulse responses ir
    (B,1|C,L), L <= T, implicitly zero-padded to T and read rotated left by ``shift`` (B,1|C) --
    the padding and roll of effects.py:86-100 are not materialised.  With ``want_peaks`` also
    returns max|x| and max|out| per (B,C) row, found inside the transforms (effects.py:160, :175).
    The length must have a four-step plan (``longconv_supported``)."""
    _require_native_ok(x)
    B, C, T = x.shape
    Cir, L = ir.shape[1], ir.shape[2]
    x = x.contiguous()
    ir = ir.contiguous()
    if scale is not None:
        scale = scale.reshape(B, Cir).to(torch.float32).contiguous()
    sh = None if shift is None else shift.reshape(B, Cir).to(torch.int64).contiguous()
    if x.data_ptr() % 8:
        x = x.clone()          # float2 reads (see fftconv)
    if ir.data_ptr() % 8:
        ir = ir.clone()
    lib = _native.lib()
    tb = tables.longconv_tables(T, x.device)
    need = int(lib.at_longconv_workspace_bytes(B, C, Cir, T))
    if need < 0:
        _native.check(need, "at_longconv_workspace_bytes")
    ws = _workspace(need, x.device)
    out = torch.empty_like(x)
    xpk = torch.empty((B, C), dtype=torch.float32, device=x.device) if want_peaks else None
    ypk = torch.empty((B, C), dtype=torch.float32, device=x.device) if want_peaks else None
    code = lib.at_longconv_room_f32(_native.ptr(x), _native.ptr(ir), L, L, _native.ptr(sh), _native.ptr(scale), B, C, Cir, T,
                                    _native.ptr(tb), _native.ptr(out), _native.ptr(xpk), _native.ptr(ypk), _native.ptr(ws),
                                    ws.numel(), _native.current_stream(x.device))
    _native.check(code, "at_longconv_room_f32")
    return (out, xpk, ypk) if want_peaks else out


# ----------------------------------------------------------- phase vocoder
def phase_vocoder(X_bcfn: torch.Tensor, p: int, q: int, hop: int) -> torch.Tensor:
    """Time-scale modification of a (B, C, F, N) complex64 HIP spectrum by the rate p/q:
    returns (B, C, F, ceil(N q / p)) in the same bin-contiguous physical layout."""
    _require_native_ok(X_bcfn)
    B, C, F, N = X_bcfn.shape
    X = X_bcfn.transpose(2, 3).contiguous()        # physical (B, C, N, F); no copy for stft() outputs
    lib = _native.lib()
    n_out = int(lib.at_phase_vocoder_frames(N, int(p), int(q)))
    if n_out < 0:
        _native.check(n_out, "at_phase_vocoder_frames")
    Y = torch.empty((B, C, n_out, F), dtype=torch.complex64, device=X.device)
    code = lib.at_phase_vocoder_f32(_native.ptr(torch.view_as_real(X)), B * C, N, F, int(p), int(q), int(hop),
                                    _native.ptr(torch.view_as_real(Y)), n_out, _native.current_stream(X.device))
    _native.check(code, "at_phase_vocoder_f32")
    return Y.transpose(2, 3)
