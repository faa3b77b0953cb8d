"""Multi-GPU execution of the hot path: one process per GPU, the batch dimension sharded in
contiguous slabs, shared read-only tables broadcast once from rank 0 over RCCL/xGMI
(``torch.distributed`` backend "nccl" IS RCCL on ROCm), no collective on the data path
(SURVEY.md 8(e)).  Every hot-path op is independent per batch item, so there is nothing to
reduce; the optional gather of per-item results is a convenience for the caller.

CPU tests drive the same code with the ``gloo`` backend (tests/test_dist_gloo.py).
"""
import os

import torch
import torch.distributed as dist

from . import tables


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


_FORCE_COLLECTIVES = False      # init(force=True): a single rank runs the collective code paths too (first contact with RCCL)


def init(backend: str = None, force: bool = False, shared_device: bool = False):
    """Join the process group described by the environment (no-op for a single process unless ``force``: then ONE rank
    initialises the group -- RCCL on a HIP device, the reference's convention audiotools/ml/accelerator.py:43-48 -- and the
    table broadcasts below go through their collective path, so that a 1-GPU box exercises everything but the wire).
    ``shared_device`` (tests only): every rank uses cuda:0 and the group runs over gloo -- RCCL refuses two ranks on one
    GPU -- so that a 1-GPU box can walk the multi-rank control flow (which rank enters which collective).
    Returns (rank, world_size, device)."""
    global _FORCE_COLLECTIVES
    rank, local_rank, world = env_world()
    use_cuda = torch.cuda.is_available()
    if shared_device:
        local_rank, backend = 0, (backend or "gloo")
    if use_cuda:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    _FORCE_COLLECTIVES = bool(force)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if use_cuda else "gloo"
        kwargs = {"device_id": device} if (backend == "nccl" and use_cuda) else {}
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous slab [lo, hi) of the batch owned by ``rank`` (sizes differ by at most 1)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def broadcast_tables(items, device, src: int = 0):
    """Build every table of ``items`` = [(key, builder), ...] (numpy array / tuple of arrays each) on ``src`` only and
    install them in the per-device table cache of every rank with TWO collectives for the whole list: one small object
    broadcast (keys' shapes / dtypes / offsets) and one broadcast of a single packed byte buffer on ``device`` (RCCL on
    HIP devices) -- not one pickled round trip + one broadcast per table.  Returns the list of device tensor(s)."""
    import numpy as np

    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1 and not (_FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized()):
        return [tables.device_table(key, device, builder) for key, builder in items]
    rank = dist.get_rank()
    meta = [None]
    packed = None
    if rank == src:
        specs, chunks, off = [], [], 0
        for _key, builder in items:
            val = builder()
            arrays = [np.ascontiguousarray(a) for a in (val if isinstance(val, tuple) else (val,))]
            entry = []
            for a in arrays:
                entry.append((tuple(a.shape), str(a.dtype), off, a.nbytes))
                chunks.append((off, a))
                off = (off + a.nbytes + 255) // 256 * 256          # every table starts on a 256-byte boundary
            specs.append((entry, isinstance(val, tuple)))
        packed = np.zeros(max(off, 256), dtype=np.uint8)
        for o, a in chunks:
            packed[o: o + a.nbytes] = a.reshape(-1).view(np.uint8)
        meta = [(specs, int(packed.size))]
    dist.broadcast_object_list(meta, src=src)
    specs, total = meta[0]
    buf = torch.from_numpy(packed).to(device) if rank == src else torch.empty(total, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src)
    out = []
    for (key, _builder), (entry, is_tuple) in zip(items, specs):
        ts = []
        for shape, dtype, off, nbytes in entry:
            t = buf[off: off + nbytes].view(getattr(torch, dtype)).reshape(shape).clone()     # own storage per table
            ts.append(t)
        val = tuple(ts) if is_tuple else ts[0]
        tables.install_table(key, device, val)
        out.append(val)
    return out


def rccl_contact(device, t_start=None):
    """First contact with the collective library on THIS box (bench.py --force-nccl, one rank or many): the communicator
    set-up time, the packed table broadcast of the north-star kernel, a 100 MB impulse-response bank (the size class of
    cfg4's shared bank) and a barrier, each timed on the host with the device drained.  Returns a dict for the bench line."""
    import time

    import numpy as np

    assert dist.is_available() and dist.is_initialized(), "rccl_contact() needs an initialised process group"
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    if device.type == "cuda":
        torch.cuda.synchronize(device)
        try:
            out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # pragma: no cover
            out["nccl_version"] = f"unavailable ({type(e).__name__})"
    if t_start is not None:
        out["init_and_first_barrier_ms"] = 1e3 * (time.perf_counter() - t_start)

    def timed(fn):
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        r = fn()
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        return r, 1e3 * (time.perf_counter() - t0)

    _, out["stft_mel_tables_broadcast_ms"] = timed(lambda: broadcast_tables(
        [((k[0] + "_contact",) + tuple(k[1:]), b) for k, b in _stft_mel_items(44100, 2048, "hann", 80)], device))
    n = 25 * 1000 * 1000                                    # 100 MB of float32
    bank, out["ir_bank_100MB_broadcast_ms"] = timed(lambda: broadcast_table(
        ("contact_ir_bank", n), lambda: (np.arange(n, dtype=np.int64) % 65521).astype(np.float32), device))
    out["ir_bank_checksum_ok"] = bool(float(bank[-1]) == float((n - 1) % 65521) and float(bank[12345]) == 12345.0
                                      and float(bank[::1000].double().sum()) == float(sum((i % 65521) for i in range(0, n, 1000))))
    out["ir_bank_GBps"] = 0.1 / (out["ir_bank_100MB_broadcast_ms"] * 1e-3)
    _, out["barrier_ms"] = timed(barrier)
    tables.drop_table(("contact_ir_bank", n), device)
    return out


def broadcast_table(key, builder, device, src: int = 0):
    """One table (see ``broadcast_tables``).  Returns the device tensor(s)."""
    return broadcast_tables([(key, builder)], device, src)[0]


def broadcast_stft_mel_tables(sample_rate, n_fft, window_type, n_mels, device, fmin=0.0, fmax=None):
    """The shared tables of the STFT+mel kernel: window, twiddles, mel unit tables (info, weights)."""
    broadcast_tables(_stft_mel_items(sample_rate, n_fft, window_type, n_mels, fmin, fmax), device)


def _stft_mel_items(sample_rate, n_fft, window_type, n_mels, fmin=0.0, fmax=None):
    return [(("window", window_type, n_fft), lambda: tables.window_np(window_type, n_fft)),
            (("stft_tw", n_fft), lambda: _twiddles_np(n_fft)),
            (("mel_units", sample_rate, n_fft, n_mels, fmin, fmax),
             lambda: tables.mel_units_np(tables.mel_filters_np(sample_rate, n_fft, n_mels, fmin, fmax)))]


def _twiddles_np(n_fft):
    import numpy as np

    from . import _native

    out = np.empty(2 * n_fft, dtype=np.float32)
    _native.check(_native.lib().at_stft_twiddles_host(n_fft, out.ctypes.data), "at_stft_twiddles_host")
    return out


def broadcast_cfg4_tables(sample_rate, n_bands, device):
    """Shared tables of the LowPass -> Equalizer -> RoomImpulseResponse chain (BASELINE configs[3]):
    the mel band-split low-pass bank the equaliser FIRs are composed from, and the 2048-point
    twiddles of the overlap-save FIR kernel.  (The IR bank itself is a caller-owned table:
    ``broadcast_table(key, builder, device)``.)"""
    def bank():
        b, _half = tables.band_split_bank(int(sample_rate), int(n_bands))
        return (b.numpy(),)

    broadcast_tables([(("band_split_bank", int(sample_rate), int(n_bands)), bank),
                      (("stft_tw", 2048), lambda: _twiddles_np(2048))], device)


def broadcast_cfg5_tables(old_sr, new_sr, n_fft, n_mels, device, window_type="hann", fmin=0.0, fmax=None):
    """Shared tables of resample -> mel_spectrogram (BASELINE configs[4]): the grouped sparse
    polyphase bank of the resampler, then window / twiddles / mel units at the NEW rate."""
    import math

    g = math.gcd(int(old_sr), int(new_sr))
    old, new = int(old_sr) // g, int(new_sr) // g

    def bank():
        wg, base = tables.resample_grouped_bank(int(old_sr), int(new_sr))[:2]
        return (wg, base)

    def mfma_bank():
        W, lo = tables.resample_mfma_bank(int(old_sr), int(new_sr))[:2]
        return (W, lo)

    def f16_bank():
        import numpy as np
        W, lo = tables.resample_f16_bank(int(old_sr), int(new_sr))[:2]
        return (W.view(np.int32), lo)

    items = []
    if old != new:
        # the bank of whichever kernel kernels.resample() dispatches to (odd reduced source rate:
        # the matrix-core forms -- fp16-split for 64..256 output phases --; otherwise the VALU form)
        from . import kernels
        form = kernels.resample_first_form(old, new)           # the same predicate the dispatcher uses
        items.append({"f16": (("resample_f16", old, new), f16_bank), "mfma": (("resample_mfma", old, new), mfma_bank),
                      "grouped": (("resample_grouped", old, new), bank)}[form])
    broadcast_tables(items + _stft_mel_items(int(new_sr), n_fft, window_type, n_mels, fmin, fmax), device)


def gather_items(local: torch.Tensor, n_items: int):
    """All-gather per-item results (e.g. the (B_local,) LUFS vector) into the full batch order."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return local
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)
