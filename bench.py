#!/usr/bin/env python
"""Benchmarks of the batched DSP hot path on 1/2/4/8 MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config north_star|cfg4|cfg5]

``--gpus N`` with N > 1 and no torchrun environment re-launches this script under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (one
process per GPU, the reference's own convention, audiotools/ml/accelerator.py:35-48); under an
existing torchrun environment (the driver's) it just joins the group.  The world size must equal
``--gpus``.  Rank 0 prints ONE JSON line.

Configurations (SURVEY.md 8(d) synthetic inputs, resident in HBM before the timed region):
  north_star  batch 512 x 2ch x 10 s @44.1 kHz:  sig.mel_spectrogram(80) [fused STFT 2048/512 hann
              + mel, stores stft_data]  +  sig.loudness() [K-weighting IIR + gated BS.1770]
              -- BASELINE.json ``metric``; this is what the driver runs.
  cfg4        batch 1024 x mono x 5 s @48 kHz:  Compose(LowPass(4/8/16 kHz), Equalizer(6 bands),
              RoomImpulseResponse(2 s RIR, DRR, EQ)) with parameters drawn beforehand
  cfg5        batch 2048 x 2ch x 30 s @44.1 kHz:  resample(16000) + mel_spectrogram(80)
Multi-GPU: the batch is sharded in contiguous slabs over the ranks (strong scaling: the metric
fixes the global batch); shared tables (window, twiddles, mel units, band-split bank, resample
bank, the IR bank) are built on rank 0 and broadcast once over RCCL; no collective on the data path.
value = global items x seconds per item / (max over ranks of the K-step time / K).

Launches are eager and asynchronous: the host enqueues step k+1 while the GPU runs step k, so
Python dispatch (~0.15 ms per step) is hidden even at 64 items per GPU (0.44 ms of kernels).
``--graph`` replays the step from a captured HIP graph instead; measured SLOWER on this stack
(0.475 vs 0.437 ms per step at 64 items, 2.86 vs 2.82 ms at 512: profiles/r02_notes.md), so it is
not the default.

JSON extras: "roofline" (algorithmic HBM bytes of the dominant kernel / its HIP-event duration vs
8 TB/s) and "cpu_baseline" (the oracle's port of the reference's CPU path -- bit-equal to the
unmodified reference for stft/mel, <= 1e-5 LU for loudness, tests/test_oracle_vs_reference.py --
timed on the host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_MELS = 80
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec

CONFIGS = {
    "north_star": dict(batch=512, ch=2, sr=44100, dur=10.0, chdesc="2ch", durdesc="10s@44.1kHz",
                       metric="audio-seconds/sec (STFT+mel+LUFS pipeline)"),
    "cfg4": dict(batch=1024, ch=1, sr=48000, dur=5.0, chdesc="mono", durdesc="5s@48kHz",
                 metric="audio-seconds/sec (LowPass->Equalizer->RoomImpulseResponse chain)"),
    "cfg5": dict(batch=2048, ch=2, sr=44100, dur=30.0, chdesc="2ch", durdesc="30s@44.1kHz",
                 metric="audio-seconds/sec (resample 44.1k->16k + STFT + mel)"),
}


def make_batch(n_items, ch, T, sr, device, seed):
    """SURVEY.md 8(d) synthetic input, generated directly on the owning device."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = (0.1 * torch.randn(n_items, ch, T, device=device, generator=g)).clamp_(-1, 1)
    gain = 10 ** (-30 * torch.rand(n_items, device=device, generator=g) / 20)
    x *= gain[:, None, None]
    if T >= 5 * sr:
        for i in range(0, n_items, 20):  # 5 % of the items: 2 s of digital silence
            x[i, :, 3 * sr: 5 * sr] = 0
    return x


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline_north_star(n_items, iters):
    """The reference's CPU path as restated by the oracle (torch CPU stft/abs/matmul + the C DF-I
    lfilter + gating: the ops the reference executes), B = 64 by default, median of >= 3 passes.
    torch's CPU kernels do not scale to very wide hosts, so a few thread counts are tried and
    the best MEDIAN is reported with its thread count."""
    from oracle import cport, restate

    cport.build()
    ncpu = os.cpu_count()
    sr, T = 44100, 441000
    x = make_batch(n_items, 2, T, sr, torch.device("cpu"), 999)
    best, best_threads, spent = None, None, 0.0
    for threads in sorted({min(ncpu, t) for t in (16, 32, 64, ncpu)}):
        torch.set_num_threads(threads)
        os.environ["OMP_NUM_THREADS"] = str(threads)
        times = []
        for _ in range(max(iters, 3)):
            t0 = time.perf_counter()
            X = restate.stft(x, 2048, 512, "hann")
            mel = restate.mel_spectrogram(X, sr, N_MELS)
            lufs = restate.loudness(x, sr)
            times.append(time.perf_counter() - t0)
            del X, mel, lufs
        spent += sum(times)
        med = statistics.median(times)
        if best is None or med < best:
            best, best_threads = med, threads
        if spent > 25.0:
            break
    return {"value": n_items * 10.0 / best, "unit": "audio-seconds/sec", "cores": best_threads,
            "kind": "port", "host_cpus": ncpu,
            "port_evidence": "oracle/restate.py is bit-equal to the unmodified reference on stft/mel and within "
                             "1e-5 LU on loudness (tests/test_oracle_vs_reference.py; /root/reference is absent here)",
            "sample": f"{n_items} items x 2ch x 10s@44.1kHz, stft(2048/512)+mel80+LUFS(IIR), median of "
                      f"{max(iters, 3)} passes ({best:.2f} s) at the best of the thread counts tried (<= {ncpu}); "
                      f"torch CPU ops + oracle/c DF-I lfilter (OpenMP)"}


def cpu_baseline_cfg(config, kw_cpu, n_items):
    """Oracle port of the other configurations on a few items (bounded CPU time)."""
    from oracle import restate

    ncpu = os.cpu_count()
    torch.set_num_threads(min(ncpu, 32))
    c = CONFIGS[config]
    sr, T = c["sr"], int(c["sr"] * c["dur"])
    x = make_batch(n_items, c["ch"], T, sr, torch.device("cpu"), 999)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        if config == "cfg4":
            k = kw_cpu["Compose"]
            y = restate.low_pass(x, k["0.LowPass"]["cutoff"][:n_items], sr)
            y = restate.equalizer(y, sr, k["1.Equalizer"]["eq"][:n_items])
            r = k["2.RoomImpulseResponse"]
            restate.apply_ir(y, r["ir_signal"].audio_data[:n_items], sr, r["drr"][:n_items], r["eq"][:n_items])
        else:
            y = restate.resample(x, sr, 16000)
            restate.mel_spectrogram(restate.stft(y, 2048, 512), 16000, N_MELS)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": n_items * c["dur"] / med, "unit": "audio-seconds/sec", "cores": min(ncpu, 32), "kind": "port",
            "host_cpus": ncpu, "sample": f"{n_items} items of {config}, median of 3 passes ({med:.2f} s), oracle/restate.py"}


def lib_sha256():
    """sha256 of the library this process loaded: the bench line and every profiles/*_pmc_summary.json carry it, so that a
    replayed counter value can be tied to the binary it was measured on."""
    import hashlib
    from audiotools_amd import _native
    try:
        return hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None


def committed_traffic(config, n_local, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS
    configuration and shape (profiles/r0N_<tag>_pmc_summary.json; FETCH_SIZE and WRITE_SIZE in separate passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no profile of this
    exact shape is committed -- it is a recorded measurement, not one made by this run.  Also returns the per-kernel
    table {kernel: bytes per launch} of the same passes (every kernel of the configuration, not only the dominant one)."""
    tag = {"north_star": "bench", "cfg4": "cfg4", "cfg5": "cfg5"}[config]
    want_items = CONFIGS[config]["batch"]
    if n_local != want_items:
        return None, None, None
    sha = lib_sha256()
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        name = f"{rnd}_{tag}_pmc_summary.json"
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
            if d.get("lib_sha256") != sha:
                # counters of ANOTHER binary say nothing about this one: traffic stays null (the newest summary decides)
                return None, (f"profiles/{name} was recorded on library sha256 {str(d.get('lib_sha256'))[:16]}, this run loaded "
                              f"{str(sha)[:16]}: not replayed"), None
            per_kernel = {}
            for k, v in d["pmc_fetch"].items():
                w = d["pmc_write"].get(k, {}).get("WRITE_SIZE")
                if w is not None and "FETCH_SIZE" in v:
                    per_kernel[k] = (2.0 * v["FETCH_SIZE"] + w) * 1024.0
            dom = [v for k, v in per_kernel.items() if kernel_substr in k]
            if not dom:
                continue
            box = d.get("box", "another MI355X of the pool (not this run's box)")
            return dom[0], (f"profiles/{name}: rocprofv3 --pmc of the same command, recorded on {box}; "
                            f"a committed measurement replayed here, NOT taken by this run"), per_kernel
        except Exception:
            continue
    return None, None, None


# ----------------------------------------------------------------------------- parity of the benchmarked outputs
def rel_rows(got, ref):
    """max over (item, channel) rows of max|got - ref| / max|ref| (the per-row measure of tests/test_gpu_parity.py)."""
    got, ref = got.double().reshape(-1, got.shape[-2] * got.shape[-1] if got.dim() > 3 else got.shape[-1]), \
        ref.double().reshape(-1, ref.shape[-2] * ref.shape[-1] if ref.dim() > 3 else ref.shape[-1])
    den = ref.abs().amax(-1).clamp_min(1e-30)
    return float(((got - ref).abs().amax(-1) / den).max())


def parity_items(n_local):
    return sorted({0, n_local // 2 + (1 if n_local > 2 else 0), n_local - 1})[:3]


def parity_north_star(x, sr, mel, stft, lufs):
    """The benchmarked launch itself (all n_local items in one launch) against the oracle on three of its items:
    north_star tolerance 1e-4 relative (per row) for stft / mel, 0.1 LU for loudness."""
    from oracle import restate
    idx = parity_items(x.shape[0])
    xc = x[idx].cpu()
    X = restate.stft(xc, 2048, 512, "hann")
    ref_mel = restate.mel_spectrogram(X, sr, N_MELS)
    ref_lufs = restate.loudness(xc, sr)
    got_stft = torch.view_as_real(stft[idx].cpu())
    out = {"items": idx, "stft_rel": rel_rows(got_stft.flatten(-2), torch.view_as_real(X).flatten(-2)),
           "mel_rel": rel_rows(mel[idx].cpu(), ref_mel),
           "lufs_abs": float((lufs[idx].cpu().double() - ref_lufs.double()).abs().max()),
           "tol": {"stft_rel": 1e-4, "mel_rel": 1e-4, "lufs_abs": 0.1}, "oracle": "oracle/restate.py on the host"}
    out["ok"] = bool(out["stft_rel"] < 1e-4 and out["mel_rel"] < 1e-4 and out["lufs_abs"] < 0.1)
    return out


def parity_device_north_star(x, sr, mel, stft, chunk=16):
    """Every rank's own check of its own outputs, on its device, without the oracle (which lives on rank 0's host): EVERY item
    of the rank's slab, in chunks, against torch.stft / |X| @ basis^T in float64 (the formulation the reference runs,
    audio_signal.py:1195-1202, 1355-1368), per-row relative error as in tests/test_gpu_parity.py.  Catches what a three-item
    check on rank 0 cannot: a table that arrived wrong on rank r, a pooled buffer aliased on rank r, one bad row out of 1024.
    Returns (ok, worst stft_rel, worst mel_rel, items checked)."""
    from audiotools_amd import tables as TB
    win = TB.window("hann", 2048, x.device).double()
    basis = torch.from_numpy(TB.mel_filters_np(sr, 2048, N_MELS, 0.0, None)).to(x.device).double()     # (n_mels, F)
    e_s = e_m = 0.0
    n = x.shape[0]
    for lo in range(0, n, chunk):
        xi = x[lo: lo + chunk].double()
        B, C, T = xi.shape
        X = torch.stft(xi.reshape(B * C, T), 2048, 512, window=win, center=True, pad_mode="reflect", return_complex=True)   # (BC, F, N)
        M = (X.abs().transpose(-1, -2) @ basis.T).transpose(-1, -2)
        got_s = stft[lo: lo + chunk].reshape(B * C, *stft.shape[2:]).to(torch.complex128)
        got_m = mel[lo: lo + chunk].reshape(B * C, *mel.shape[2:]).double()
        den_s = X.abs().amax((-1, -2)).clamp_min(1e-30)
        den_m = M.abs().amax((-1, -2)).clamp_min(1e-30)
        e_s = max(e_s, float(((got_s - X).abs().amax((-1, -2)) / den_s).max()))
        e_m = max(e_m, float(((got_m - M).abs().amax((-1, -2)) / den_m).max()))
        del X, M, got_s, got_m, xi
    return bool(e_s < 1e-4 and e_m < 1e-4), e_s, e_m, n


# ----------------------------------------------------------------------------- launching
def respawn_under_torchrun(n):
    """python bench.py --gpus N without a torchrun environment: start N ranks on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, adist, rank, world, device, rccl_contact=None):
    """Everything around the kernels: one process per rank, the world size check, contiguous batch
    slabs, the table broadcasts of the chosen configuration (gloo when there is no GPU)."""
    cfg = CONFIGS[args.config]
    batch = cfg["batch"] if args.batch is None else args.batch
    lo, hi = adist.shard_range(batch, rank, world)
    if args.config == "north_star":
        adist.broadcast_stft_mel_tables(cfg["sr"], 2048, "hann", N_MELS, device)
    elif args.config == "cfg4":
        import numpy as np
        adist.broadcast_table(("bench_ir_bank", cfg["sr"]), lambda: np.arange(12, dtype=np.float32).reshape(3, 1, 4), device)
        adist.broadcast_cfg4_tables(cfg["sr"], 6, device)
    else:
        adist.broadcast_cfg5_tables(cfg["sr"], 16000, 2048, N_MELS, device)
    sizes = torch.tensor([hi - lo], dtype=torch.int64, device=device)
    if world > 1:
        torch.distributed.all_reduce(sizes)
    assert int(sizes[0]) == batch, "the slabs must cover the batch exactly once"
    if rank == 0:
        print(json.dumps({"dry_run": True, "config": args.config, "n_gpus": world, "global_batch": batch,
                          "items_rank0": hi - lo, "rccl_contact": rccl_contact,
                          "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def time_steps(step, steps, warmup, adist, device, world, n_events=0):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; the
    elapsed time is the MAX over ranks."""
    if not _PRIMED.get(id(step)):
        # set-up, not warm-up: the first call of a shape builds tables and allocates workspaces -- kept out of the timed K
        # steps even when the driver asks for --warmup 0
        step(None)
        _PRIMED[id(step)] = True
    for _ in range(warmup):
        step(None)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(n_events)] for _ in range(steps)] if n_events else None
    adist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(events[k] if events else None)
    torch.cuda.synchronize()
    adist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    PER_RANK_S[:] = [t1 - t0]
    if world > 1:
        every = [torch.zeros_like(elapsed) for _ in range(world)]
        torch.distributed.all_gather(every, elapsed)
        PER_RANK_S[:] = [float(e[0]) for e in every]
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    return float(elapsed[0]), events


PER_RANK_S = []     # wall time of the last timed region on every rank (rank order)
_PRIMED = {}        # step functions that have had their one set-up call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="north_star", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="GLOBAL batch (items); default: the configuration's")
    ap.add_argument("--cpu-items", type=int, default=None)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the north-star step from a captured HIP graph instead of eager launches")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the floor twin and the plain-allocation probe behind the timed region (the --stats pass of "
                         "tools/profile_round.sh: its per-kernel average should hold the timed launches, not the probes')")
    ap.add_argument("--no-share", action="store_true",
                    help="skip the share_64 block (the 8-GPU share timed on this device): profile runs use it so that the "
                         "per-kernel averages of rocprofv3 only see launches of the benchmark's own size")
    ap.add_argument("--placement", choices=("on", "off"), default="on",
                    help="north star: opt into the library's placement-aware output pool (kernels.output_placement: the real kernel "
                         "timed into up to twelve candidate spectrum buffers once, the fastest kept; OFF by default in the library, an "
                         "explicit opt-in here) for the timed region.  The same K steps on plain torch.empty allocations are timed "
                         "right after it and printed as roofline.placement.*_plain_allocation either way")
    ap.add_argument("--force-nccl", action="store_true",
                    help="initialise the RCCL process group even for ONE rank and run the multi-GPU set-up path on it (table "
                         "broadcasts, a 100 MB impulse-response bank broadcast, barriers): first contact with librccl on a "
                         "1-GPU box, reported as \"rccl_contact\" (VERDICT r05 #2)")
    ap.add_argument("--shared-device", action="store_true",
                    help="TEST ONLY: all ranks on cuda:0 over gloo (RCCL refuses two ranks on one GPU): walks the multi-rank "
                         "control flow -- which rank enters which collective -- on a 1-GPU box; the numbers mean nothing")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only (process group, sharding, table broadcast; gloo on a CPU-only host): no kernels")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))

    import audiotools_amd as A
    from audiotools_amd import _native, dist as adist

    # RCCL's own warnings go to stderr from the start: the first multi-GPU run is the driver's, not ours
    os.environ.setdefault("NCCL_DEBUG", "VERSION" if args.force_nccl else "WARN")   # (VERSION: RCCL's banner on stderr, kept with the log)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (the host driver has no legacy IPC)
    rccl_contact = None
    try:
        t_init = time.perf_counter()
        rank, world, device = adist.init(force=args.force_nccl, shared_device=args.shared_device)
        adist.barrier()                                            # first collective: RCCL communicator set-up over xGMI
        if args.force_nccl:
            rccl_contact = adist.rccl_contact(device, t_init)
    except Exception as e:                                         # pragma: no cover - needs a broken fabric
        if int(os.environ.get("RANK", 0)) == 0:
            print(json.dumps({"error": f"process group / first barrier failed: {type(e).__name__}: {e}",
                              "n_gpus": args.gpus, "WORLD_SIZE": os.environ.get("WORLD_SIZE"),
                              "NCCL_DEBUG": os.environ.get("NCCL_DEBUG"),
                              "hint": "RCCL warnings of every rank are on stderr (NCCL_DEBUG=WARN)"}))
        raise
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s) "
                         f"(WORLD_SIZE={os.environ.get('WORLD_SIZE')})")
    if int(os.environ.get("LOCAL_RANK", 0)) == 0:
        _native.build()   # no-op when the in-tree library is up to date (it normally travels prebuilt)
    adist.barrier()
    if args.dry_run:
        return dry_run(args, adist, rank, world, device, rccl_contact)
    assert device.type == "cuda", "bench.py needs a GPU (the product has no CPU fallback)"
    if world > 1 and not args.shared_device:
        assert torch.distributed.get_backend() == "nccl", "multi-GPU runs use RCCL (torch backend 'nccl')"

    cfg = CONFIGS[args.config]
    batch = cfg["batch"] if args.batch is None else args.batch
    sr, ch, dur = cfg["sr"], cfg["ch"], cfg["dur"]
    T = int(sr * dur)
    lo, hi = adist.shard_range(batch, rank, world)
    n_local = hi - lo
    x = make_batch(n_local, ch, T, sr, device, 1234 + rank)
    rows = n_local * ch
    out = {"metric": f"{cfg['metric']}, batch {batch}x{cfg['chdesc']}x{cfg['durdesc']}", "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "setup": "one untimed set-up call before the W warm-up steps (tables, workspaces, with --placement on the calibration of the output pool)"}
    kw_cpu = None

    if args.config == "north_star":
        sig = A.AudioSignal(x, sr)
        n_fft, hop = sig.stft_params.window_length, sig.stft_params.hop_length
        adist.broadcast_stft_mel_tables(sr, n_fft, "hann", N_MELS, device)
        n_frames, F = 1 + T // hop, n_fft // 2 + 1
        stft_bytes = rows * T * 4 + rows * n_frames * F * 8 + rows * n_frames * N_MELS * 4
        lufs_bytes = rows * T * 4 + n_local * 4
        res = {}

        def api_step():
            res["mel"] = sig.mel_spectrogram(N_MELS)
            sig._loudness = None
            res["lufs"] = sig.loudness()

        # per-kernel durations (HIP events on the launch stream), eager launches
        def step_events(ev):
            if ev is not None:
                ev[0].record()
            res["mel"] = sig.mel_spectrogram(N_MELS)
            if ev is not None:
                ev[1].record()
            sig._loudness = None
            res["lufs"] = sig.loudness()
            if ev is not None:
                ev[2].record()

        from audiotools_amd import kernels as K
        if args.placement == "on":
            # the library's opt-in: calibrated during the set-up call below, outside the timed region; its cost
            # (one buffer of the spectrum's size pinned, ~37 launches + one synchronisation once) is printed with the line
            # (eighteen candidates where the free memory holds them inside the pool's half-of-free limit: on boxes where fast
            #  placements are rare -- one in twelve in some calibration lists -- six more draws are worth their 40 ms, once)
            K.output_placement(enabled=True, calibrate_after=1, candidates=18)
        launch = "eager"
        step = step_events
        if args.graph:
            # (a captured graph cannot carry the per-kernel events: they are taken from a separate eager pass)
            _, events = time_steps(step_events, max(5, min(args.steps, 20)), args.warmup, adist, device, world, n_events=3)
            try:
                graph = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    api_step()
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(graph):
                    api_step()
                step = lambda ev: graph.replay()
                launch = "hipGraph replay of the captured API calls (mel_spectrogram + loudness)"
            except Exception as e:  # pragma: no cover - capture not supported on this stack
                launch = f"eager (graph capture failed: {type(e).__name__})"
                step = lambda ev: api_step()
        if args.graph:
            elapsed, _ = time_steps(step, args.steps, args.warmup, adist, device, world)
            timing_note = "HIP events around eager launches in a separate pass (the timed region replays a graph)"
        else:
            # the per-kernel HIP events are recorded INSIDE the timed region: the K launches whose durations are
            # averaged below are the K launches of ms_per_step (so stft_mel + lufs_total <= ms_per_step)
            elapsed, events = time_steps(step, args.steps, args.warmup, adist, device, world, n_events=3)
            timing_note = "HIP events on the launch stream, recorded inside the timed region (the same K steps)"
        stft_ms = sum(e[0].elapsed_time(e[1]) for e in events) / len(events)
        lufs_ms = sum(e[1].elapsed_time(e[2]) for e in events) / len(events)
        mel, lufs = res["mel"], res["lufs"]
        assert torch.isfinite(mel).all() and torch.isfinite(lufs).all()
        # the 8-GPU share timed on THIS device: what one rank of an 8-way batch shard would run, so that the
        # first measured scaling curve has a prediction to be checked against (no collective on the data path)
        share = None
        if world == 1 and n_local >= 16 and not args.graph and not args.no_share:
            n_sh = n_local // 8
            sig_sh = A.AudioSignal(x[:n_sh], sr)
            res_sh = {}

            def share_step(ev):
                res_sh["mel"] = sig_sh.mel_spectrogram(N_MELS)
                sig_sh._loudness = None
                res_sh["lufs"] = sig_sh.loudness()

            saved = list(PER_RANK_S)
            k_sh = max(args.steps, 100)          # a 0.36 ms step: 20 of them are shorter than the synchronisation around them
            el_sh, _ = time_steps(share_step, k_sh, max(args.warmup, 10), adist, device, world)
            PER_RANK_S[:] = saved
            ms_sh = 1e3 * el_sh / k_sh
            share = {"items": n_sh, "steps": k_sh, "ms_per_step": ms_sh,
                     "predicted_speedup_8gpu": (1e3 * elapsed / args.steps) / ms_sh,
                     "predicted_value_8gpu": batch * dur / (ms_sh * 1e-3),
                     "note": "one rank's share of an 8-way batch shard, timed on this device; the data path has no "
                             "collective, so 8 ranks are predicted to finish in this time (+ barrier skew)"}
        achieved = stft_bytes / (stft_ms * 1e-3) / 1e9
        traffic, traffic_src, _ = committed_traffic(args.config, n_local, "stft_mel_kernel")
        # evidence outside the timed region (parity of the benchmarked launch, the floor twin, the copy rate) is taken on
        # rank 0 only: the other ranks neither build nor run the oracle, they wait at the closing barrier
        parity = parity_north_star(x, sr, mel, sig.stft_data, lufs) if rank == 0 else {"ok": True}
        # ... and EVERY rank checks every item of its own slab on its own device (torch.stft / |X| @ basis in float64); the
        # verdicts are all-reduced, so a wrong result on any rank fails the line
        dev_ok, dev_es, dev_em, dev_n = parity_device_north_star(x, sr, mel, sig.stft_data)
        flag = torch.tensor([1 if dev_ok else 0], dtype=torch.int32, device=device)
        if world > 1:
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        parity["every_rank_device_check"] = {"ok_all_ranks": bool(int(flag[0])), "rank0_stft_rel": dev_es, "rank0_mel_rel": dev_em,
                                             "rank0_items_checked": dev_n,
                                             "what": "EVERY item of every rank's slab vs torch.stft + |X| @ mel basis (float64) on the rank's "
                                                     "device, worst per-row relative error"}
        parity["ok"] = bool(parity["ok"] and int(flag[0]))
        # the zero-compute floor of the dominant kernel's traffic on THIS box: its measurement twin (same grid, schedule,
        # addresses, load / store instructions and cache policy, no transform: at_stft_mel_floor_f32), timed the same way
        floor_ms = floor_iso_ms = floor_same_ms = None
        placement = stft_only = None
        try:
            if rank != 0 or args.no_probes:
                raise StopIteration
            from audiotools_amd import tables as TB
            win = TB.window("hann", n_fft, device)
            units = TB.mel_units(sr, n_fft, N_MELS, 0.0, None, device)
            fl_stft = torch.empty((n_local, ch, n_frames, F), dtype=torch.complex64, device=device)
            fl_mel = torch.empty((n_local, ch, n_frames, N_MELS), dtype=torch.float32, device=device)
            if K.stft_mel_floor(x, win, n_fft, hop, fl_stft, (units[0], units[1], N_MELS), fl_mel):
                for _ in range(3):
                    K.stft_mel_floor(x, win, n_fft, hop, fl_stft, (units[0], units[1], N_MELS), fl_mel)
                fe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                fe[0].record()
                n_probe = max(5, min(args.steps, 20))          # launches per probe
                for _ in range(n_probe):
                    K.stft_mel_floor(x, win, n_fft, hop, fl_stft, (units[0], units[1], N_MELS), fl_mel)
                fe[1].record()
                torch.cuda.synchronize()
                floor_ms = fe[0].elapsed_time(fe[1]) / n_probe
                # ... and one launch at a time with the device idle before each (what a kernel trace sees): on some boxes the
                # twin runs faster this way than back to back (1.73 against 2.09 ms, profiles/r04_notes.md), the real kernel
                # does not; the smaller of the two is the floor the kernel is held against
                iso = []
                for _ in range(min(args.steps, 10)):
                    torch.cuda.synchronize()
                    ie = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                    ie[0].record()
                    K.stft_mel_floor(x, win, n_fft, hop, fl_stft, (units[0], units[1], N_MELS), fl_mel)
                    ie[1].record()
                    torch.cuda.synchronize()
                    iso.append(ie[0].elapsed_time(ie[1]))
                floor_iso_ms = sum(iso) / len(iso)
                # ... and on the buffers the KERNEL of the last timed step wrote (parity has been checked above; the twin leaves
                # garbage in them): the twin's time is a property of where the 7.2 GB spectrum buffer lies physically
                # (profiles/r05_notes.md section 1: 1.6-1.7 ms on some allocations, 2.0-2.1 ms on others, deterministic per
                # allocation), so only this figure says which regime the kernel of record ran in
                own_stft, own_mel = sig.stft_data, mel
                for _ in range(2):
                    K.stft_mel_floor(x, win, n_fft, hop, own_stft, (units[0], units[1], N_MELS), own_mel)
                oe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                oe[0].record()
                for _ in range(n_probe):
                    K.stft_mel_floor(x, win, n_fft, hop, own_stft, (units[0], units[1], N_MELS), own_mel)
                oe[1].record()
                torch.cuda.synchronize()
                floor_same_ms = oe[0].elapsed_time(oe[1]) / n_probe
            # The transform without its mel stage on the same fresh buffers (north_star's gate names "the STFT kernel"): additive
            # evidence, the kernel of record stays the fused one
            stft_only = None
            try:
                for _ in range(3):
                    K.stft_mel(x, win, n_fft, hop, out=(fl_stft, None))
                se = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                se[0].record()
                n_probe = max(5, min(args.steps, 20))
                for _ in range(n_probe):
                    K.stft_mel(x, win, n_fft, hop, out=(fl_stft, None))
                se[1].record()
                torch.cuda.synchronize()
                so_ms = se[0].elapsed_time(se[1]) / n_probe
                so_bytes = rows * T * 4 + rows * n_frames * F * 8
                stft_only = {"kernel": "stft_mel_kernel_v2<0> (the same transform, no mel stage)", "avg_launch_ms": so_ms,
                             "algorithmic_bytes_per_launch": so_bytes, "frac": so_bytes / (so_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            except Exception as e:  # pragma: no cover
                stft_only = {"error": f"{type(e).__name__}: {e}"}
            del fl_stft, fl_mel
        except StopIteration:
            pass
        except Exception as e:  # pragma: no cover - the floor is evidence, never a reason to lose the line
            floor_ms = floor_iso_ms = floor_same_ms = None
            out["floor_error"] = f"{type(e).__name__}: {e}"
        # Placement (profiles/r05_notes.md section 1; kernels._PlacedOutputs, an OPT-IN since round 6).  With --placement on
        # (default) the timed region above ran on the buffer the library chose; the same K steps on the buffers torch.empty
        # hands out -- the library's default behaviour -- are timed here, and both figures are printed.  EVERY rank takes part
        # (time_steps is collective: barrier + max over ranks), after rank 0's probes above.
        if not args.no_probes and not args.graph:
            try:
                pool_rep, pool_bytes = K.output_placement(), K._placed_outputs.bytes_held()
                sig._release_stft_data()
                res.clear()
                del mel
                K.output_placement(enabled=False)               # (drops the pooled buffer)
                saved_ranks = list(PER_RANK_S)
                pl_el, pl_ev = time_steps(step_events, args.steps, max(args.warmup, 2), adist, device, world, n_events=3)
                PER_RANK_S[:] = saved_ranks                      # (per_rank_ms_per_step reports the timed region of record)
                k_plain = sum(e[0].elapsed_time(e[1]) for e in pl_ev) / len(pl_ev)
                placement = {"enabled_for_the_timed_region": args.placement == "on",
                             "pool": pool_rep, "pool_bytes_held": pool_bytes,
                             "kernel_ms_plain_allocation": k_plain,
                             "frac_plain_allocation": stft_bytes / (k_plain * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "ms_per_step_plain_allocation": 1e3 * pl_el / args.steps,
                             "value_plain_allocation": batch * dur / (pl_el / args.steps),
                             "note": "the pool is OFF by default in the library; bench.py opts in (--placement on) the way a job "
                                     "with HBM to spare would: kernels.output_placement(enabled=True) -- the library timed the real "
                                     "kernel into the candidate buffers listed under pool.calibration_ms during set-up and keeps the "
                                     "fastest (pool_bytes_held pinned until release_workspaces()).  *_plain_allocation: the same K "
                                     "steps with the pool off, on whatever torch.empty handed out in this process (a draw: "
                                     "53-60 % of 8 TB/s for the same binary, deterministic per allocation); rank 0's figures"}
                mel = res["mel"]
            except Exception as e:  # pragma: no cover
                placement = {"error": f"{type(e).__name__}: {e}"}
                if world > 1:
                    raise           # a rank that leaves the collective pass would hang the others: fail loudly instead
                mel = sig.mel_spectrogram(N_MELS)
        K.output_placement(enabled=False, calibrate_after=3, candidates=12)
        # what a plain device copy reaches on THIS box (torch.Tensor.copy_, read + write counted),
        # measured the same way: the practical ceiling next to the 8 TB/s spec
        cp_src, cp_dst = x.view(-1), torch.empty_like(x).view(-1)
        cp_dst.copy_(cp_src)
        ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ce[0].record()
        for _ in range(5):
            cp_dst.copy_(cp_src)
        ce[1].record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * cp_src.numel() * 4 / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
        del cp_dst
        out["config"] = {"workload": f"north-star: batch={batch} 2ch 10s@44.1kHz mel_spectrogram(80) "
                                     f"[fused STFT {n_fft}/{hop} hann + mel] + loudness()",
                         "global_batch": batch, "items_per_gpu": n_local, "parallelism": f"batch-shard x{world}",
                         "inputs": "device-resident (H2D excluded)", "launch": launch}
        out["roofline"] = {"bound": "hbm", "kernel": "stft_mel_kernel_v2<4> (fused STFT 2048/512 + 80-band mel)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": stft_bytes, "avg_launch_ms": stft_ms,
                           "device_copy_GBps": copy_gbs, "frac_of_device_copy": achieved / copy_gbs,
                           "floor_ms": floor_ms, "floor_ms_one_at_a_time": floor_iso_ms, "floor_ms_same_buffers": floor_same_ms,
                           "frac_of_floor_same_buffers": (floor_same_ms / stft_ms) if floor_same_ms else None,
                           "frac_of_floor": (min(floor_ms, floor_iso_ms) / stft_ms) if floor_ms else None,
                           "floor_frac_of_peak": (stft_bytes / (min(floor_ms, floor_iso_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS)
                           if floor_ms else None,
                           "placement": placement, "stft_only": stft_only,
                           "floor_note": "floor_ms = the same launch with the transform removed (stft_mel_kernel_v2<.., FLOOR>: "
                                         "identical grid, runs, addresses, load/store instructions, nt policy), HIP events: K launches back to "
                                         "back (floor_ms) and one at a time on an idle device (floor_ms_one_at_a_time), both into FRESH buffers; frac_of_floor "
                                         "uses the smaller.  floor_ms_same_buffers = the twin on the output buffers of the last timed step: "
                                         "the twin's time depends on where the spectrum buffer lies physically (1.6-1.7 or 2.0-2.1 ms, "
                                         "deterministic per allocation, profiles/r05_notes.md section 1), so this is the figure that says "
                                         "which regime the kernel of record ran in"}
        out["parity_check"] = parity
        per_step = sorted(e[0].elapsed_time(e[1]) for e in events)
        out["kernels_ms"] = {"stft_mel": stft_ms, "stft_mel_min": per_step[0], "stft_mel_median": per_step[len(per_step) // 2],
                             "stft_mel_max": per_step[-1],
                             "lufs_total": lufs_ms, "lufs_GBps": lufs_bytes / (lufs_ms * 1e-3) / 1e9,
                             "timing": timing_note}
        if share is not None:
            out["share_64" if n_local == 512 else f"share_{n_local // 8}"] = share

    elif args.config == "cfg4":
        from audiotools_amd import transforms as tfm

        # rank 0 owns the impulse-response bank and broadcasts it (RCCL); every rank then draws its
        # own items' parameters on the host (the reference's DataLoader-worker job, not timed)
        def make_bank():
            g = torch.Generator().manual_seed(77)
            t_ir = torch.arange(2 * sr) / sr
            return (torch.randn(64, 1, 2 * sr, generator=g) * torch.exp(-t_ir / 0.3)).numpy()

        bank = adist.broadcast_table(("bench_ir_bank", sr), make_bank, device)
        adist.broadcast_cfg4_tables(sr, 6, device)
        chain = tfm.Compose(tfm.LowPass(cutoff=("choice", [4000, 8000, 16000])), tfm.Equalizer(n_bands=6),
                            tfm.RoomImpulseResponse(loader=tfm.TensorLoader(bank.cpu(), sr), duration=2.0, offset=0.0))
        proto = A.AudioSignal(torch.zeros(n_local, 1, 8), sr)
        torch.set_num_threads(min(8, os.cpu_count() or 8))   # per-item host draws: tiny tensors, no wide OpenMP teams
        t0 = time.perf_counter()
        kw_cpu = chain.batch_instantiate([1000 + lo + i for i in range(n_local)], proto)
        inst_ms = (time.perf_counter() - t0) * 1e3
        n_cpu = args.cpu_items or 8
        ir_cpu = kw_cpu["Compose"]["2.RoomImpulseResponse"]["ir_signal"].audio_data[:n_cpu].clone()
        kw = A.util.prepare_batch(kw_cpu, device)     # (moves the AudioSignals of kw_cpu in place)
        kw_cpu = {"Compose": {k: ({kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                  if isinstance(v, dict) else v) for k, v in kw_cpu["Compose"].items()}}
        kw_cpu["Compose"]["2.RoomImpulseResponse"]["ir_signal"] = A.AudioSignal(ir_cpu, sr)
        res = {}

        def step(ev):
            res["y"] = chain(A.AudioSignal(x, sr), **kw).audio_data

        elapsed, _ = time_steps(step, args.steps, args.warmup, adist, device, world)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(None)
        enq_ms = (time.perf_counter() - t0) * 1e3          # host time to enqueue one chain (no sync inside)
        torch.cuda.synchronize()
        assert torch.isfinite(res["y"]).all()
        chain_bytes = 3 * 2 * rows * T * 4 + rows * 2 * sr * 4
        achieved = chain_bytes / (elapsed / args.steps) / 1e9
        # parity of the benchmarked launch: the oracle's chain on three of its items (those whose impulse responses
        # were kept on the host), per-row relative error, tolerance 1e-4
        if rank == 0:          # (rank 0 only: the other ranks never touch the oracle)
            from oracle import restate
            idx = [i for i in (0, 3, n_cpu - 1) if i < n_local]
            k4 = kw_cpu["Compose"]
            r4 = k4["2.RoomImpulseResponse"]
            yo = restate.low_pass(x[idx].cpu(), k4["0.LowPass"]["cutoff"][idx], sr)
            yo = restate.equalizer(yo, sr, k4["1.Equalizer"]["eq"][idx])
            yo = restate.apply_ir(yo, r4["ir_signal"].audio_data[idx], sr, r4["drr"][idx], r4["eq"][idx])
            c4 = rel_rows(res["y"][idx].cpu(), yo)
            out["parity_check"] = {"items": idx, "chain_rel": c4, "tol": {"chain_rel": 1e-4}, "ok": bool(c4 < 1e-4),
                                   "oracle": "oracle/restate.py low_pass -> equalizer -> apply_ir on the host"}
        traffic, traffic_src, per_kernel = committed_traffic(args.config, n_local, "rowconv_kernel")
        out["config"] = {"workload": f"cfg4: batch={batch} mono 5s@48kHz Compose(LowPass, Equalizer(6), "
                                     f"RoomImpulseResponse(2 s RIR, DRR, EQ))", "global_batch": batch,
                         "items_per_gpu": n_local, "parallelism": f"batch-shard x{world} + RCCL broadcast of the IR bank / tables",
                         "inputs": "device-resident; transform parameters drawn before the timed region",
                         "host_instantiate_ms": inst_ms, "host_enqueue_ms": enq_ms}
        out["roofline"] = {"bound": "hbm", "kernel": "whole chain (fir_fft x3, alter_drr, absmax, roll_pad, four-step FFT convolution: colfft x3 + rowconv)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": None, "algorithmic_bytes_per_launch": chain_bytes,
                           "avg_launch_ms": 1e3 * elapsed / args.steps,
                           "traffic_per_kernel_launch": per_kernel, "traffic_source": traffic_src,
                           "traffic_note": "roofline.traffic stays null for the CHAIN (17 launches of 9 kernels per step); "
                                           "traffic_per_kernel_launch holds the committed FETCH x 2 + WRITE bytes of every kernel of it"}

    else:  # cfg5
        adist.broadcast_cfg5_tables(sr, 16000, 2048, N_MELS, device)
        res = {}

        def step(ev):
            s = A.AudioSignal(x, sr)
            if ev is not None:
                ev[0].record()
            s.resample(16000)
            res["y"] = s.audio_data
            if ev is not None:
                ev[1].record()
            res["mel"] = s.mel_spectrogram(N_MELS)
            if ev is not None:
                ev[2].record()

        _, events = time_steps(step, max(3, min(args.steps, 10)), args.warmup, adist, device, world, n_events=3)
        rs_ms = sum(e[0].elapsed_time(e[1]) for e in events) / len(events)
        mel_ms = sum(e[1].elapsed_time(e[2]) for e in events) / len(events)
        elapsed, _ = time_steps(step, args.steps, args.warmup, adist, device, world)
        assert torch.isfinite(res["mel"]).all()
        T2 = int(16000 * T // sr)
        n_frames = 1 + T2 // 512
        rs_bytes = rows * T * 4 + rows * T2 * 4
        mel_bytes = rows * T2 * 4 + rows * n_frames * 1025 * 8 + rows * n_frames * N_MELS * 4
        achieved = rs_bytes / (rs_ms * 1e-3) / 1e9
        if rank == 0:          # (rank 0 only: the other ranks never touch the oracle)
            from oracle import restate
            idx = sorted({0, n_local - 1})
            yo = restate.resample(x[idx].cpu(), sr, 16000)
            mo = restate.mel_spectrogram(restate.stft(yo, 2048, 512), 16000, N_MELS)
            e_rs, e_mel = rel_rows(res["y"][idx].cpu(), yo), rel_rows(res["mel"][idx].cpu(), mo)
            out["parity_check"] = {"items": idx, "resample_rel": e_rs, "mel_rel": e_mel, "tol": {"resample_rel": 1e-4, "mel_rel": 1e-4},
                                   "ok": bool(e_rs < 1e-4 and e_mel < 1e-4), "oracle": "oracle/restate.py resample -> stft -> mel on the host"}
        out["config"] = {"workload": f"cfg5: batch={batch} 2ch 30s@44.1kHz resample(16000) + mel_spectrogram(80) "
                                     f"[STFT 2048/512 as the signal keeps its stft_params]", "global_batch": batch,
                         "items_per_gpu": n_local, "parallelism": f"batch-shard x{world} + RCCL broadcast of the resample bank / tables",
                         "inputs": "device-resident (H2D excluded)"}
        from audiotools_amd import kernels as K5
        form = K5.resample_first_form(441, 160)            # the dispatcher's own predicate
        if form == "f16":
            rs_sub = "resample_f16s_rp_kernel"
            rs_kernel = rs_sub + " (banded-GEMM polyphase 441->160, fp16-split products on v_mfma_f32_16x16x32_f16)"
        elif form == "mfma":
            rs_kernel, rs_sub = "resample_mfma_ws_kernel (banded-GEMM polyphase 441->160 on v_mfma_f32_16x16x4_f32)", "resample_mfma_ws_kernel"
        else:
            rs_kernel, rs_sub = "resample_kernel (sparse polyphase 441->160)", "resample_kernel"
        traffic, traffic_src, per_kernel = committed_traffic(args.config, n_local, rs_sub)
        out["roofline"] = {"bound": "hbm", "kernel": rs_kernel, "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_source": traffic_src, "traffic_per_kernel_launch": per_kernel,
                           "algorithmic_bytes_per_launch": rs_bytes, "avg_launch_ms": rs_ms}
        out["kernels_ms"] = {"resample": rs_ms, "stft_mel": mel_ms, "stft_mel_GBps": mel_bytes / (mel_ms * 1e-3) / 1e9}
        if rank == 0:
            # "stft_params stay (2048, 512) after resample -- measure that literal behaviour; additionally report
            # (512, 128)", the parameters a 16 kHz signal would get.  Outside the timed region, rank 0 only.
            s16 = A.AudioSignal(res["y"], 16000)
            res.pop("mel", None)
            for _ in range(2):
                s16.mel_spectrogram(N_MELS, window_length=512, hop_length=128)
            ev5 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev5[0].record()
            for _ in range(3):
                s16.mel_spectrogram(N_MELS, window_length=512, hop_length=128)
            ev5[1].record()
            torch.cuda.synchronize(device)
            ms5 = ev5[0].elapsed_time(ev5[1]) / 3
            nf5 = 1 + T2 // 128
            b5 = rows * T2 * 4 + rows * nf5 * 257 * 8 + rows * nf5 * N_MELS * 4
            out["kernels_ms"]["stft_mel_512_128"] = ms5
            out["kernels_ms"]["stft_mel_512_128_GBps"] = b5 / (ms5 * 1e-3) / 1e9
            del s16

    if rank == 0:
        out["ms_per_step"] = 1e3 * elapsed / args.steps
        out["value"] = batch * dur / (elapsed / args.steps)
        out["world_size"] = world
        out["backend"] = torch.distributed.get_backend() if (world > 1 or args.force_nccl) else None
        out["per_rank_ms_per_step"] = [1e3 * t / args.steps for t in PER_RANK_S]
        if PER_RANK_S:
            out["rank_skew"] = {"min_ms_per_step": 1e3 * min(PER_RANK_S) / args.steps, "max_ms_per_step": 1e3 * max(PER_RANK_S) / args.steps,
                                "max_over_min": max(PER_RANK_S) / min(PER_RANK_S)}
        out["host"] = socket.gethostname()
        out["lib_sha256"] = lib_sha256()
        if rccl_contact is not None:
            out["rccl_contact"] = rccl_contact
        if world == 1 and not args.no_cpu_baseline:
            if args.config == "north_star":
                out["cpu_baseline"] = cpu_baseline_north_star(args.cpu_items or 64, args.cpu_iters)
            else:
                out["cpu_baseline"] = cpu_baseline_cfg(args.config, kw_cpu, args.cpu_items or (8 if args.config == "cfg4" else 4))
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1 or args.force_nccl:
        adist.barrier()            # the other ranks wait here while rank 0 takes its evidence and prints the line
        torch.distributed.destroy_process_group()
    if rank == 0 and not out.get("parity_check", {"ok": True})["ok"]:
        sys.exit("bench.py: the benchmarked outputs differ from the oracle beyond the tolerance (parity_check above)")


if __name__ == "__main__":
    main()
