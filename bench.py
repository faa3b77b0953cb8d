#!/usr/bin/env python
"""Headline benchmark: audio-seconds/sec of the STFT+mel+LUFS pipeline on a
batch of 512 x 2 ch x 10 s @ 44.1 kHz (BASELINE.json `metric`), 1/2/4/8 GPUs.

One "step" = one pass of the hot path over the whole synthetic batch, inputs resident in
HBM:   sig.mel_spectrogram(80)   -> fused HIP STFT (2048/512 hann, stores stft_data) + mel
       sig.loudness()            -> HIP K-weighting IIR + gated BS.1770 integration
Multi-GPU: the 512 items are sharded in contiguous slabs over the ranks (strong scaling, the
total work is fixed as BASELINE.json's metric states), tables broadcast once over RCCL, no
collective on the data path.  value = 512 items * 10 s / (max over ranks of the time of K
steps / K).

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  "roofline":     the fused STFT+mel kernel, algorithmic bytes / HIP-event duration vs 8 TB/s
  "cpu_baseline": the oracle port (torch CPU stft/abs/matmul + C DF-I lfilter + gating, i.e.
                  the ops the reference's CPU path executes) timed on the host cores on a
                  bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

SR = 44100
DUR = 10.0
CH = 2
N_MELS = 80
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def make_batch(n_items, device, seed):
    """SURVEY.md 8(d) synthetic input, generated directly on the owning device."""
    T = int(SR * DUR)
    g = torch.Generator(device=device).manual_seed(seed)
    x = (0.1 * torch.randn(n_items, CH, T, device=device, generator=g)).clamp_(-1, 1)
    gain = 10 ** (-30 * torch.rand(n_items, device=device, generator=g) / 20)
    x *= gain[:, None, None]
    for i in range(0, n_items, 20):  # 5 % of the items: 2 s of digital silence
        x[i, :, 3 * SR: 5 * SR] = 0
    return x


def cpu_baseline(n_items, iters):
    """Reference CPU path restated (oracle port) on the host cores.  torch's CPU kernels do not
    scale to very wide hosts, so a few thread counts are tried and the FASTEST is reported
    (`cores` = the thread count that won)."""
    from oracle import cport, restate

    cport.build()
    ncpu = os.cpu_count()
    x = make_batch(n_items, torch.device("cpu"), 999)
    best, best_threads, spent = None, None, 0.0
    for threads in sorted({min(ncpu, t) for t in (16, 32, 64, ncpu)}):
        torch.set_num_threads(threads)
        os.environ["OMP_NUM_THREADS"] = str(threads)
        for _ in range(iters):
            t0 = time.perf_counter()
            X = restate.stft(x, 2048, 512, "hann")
            mel = restate.mel_spectrogram(X, SR, N_MELS)
            lufs = restate.loudness(x, SR)
            dt = time.perf_counter() - t0
            spent += dt
            del X, mel, lufs
            if best is None or dt < best:
                best, best_threads = dt, threads
        if spent > 25.0:
            break
    return {"value": n_items * DUR / best, "unit": "audio-seconds/sec", "cores": best_threads, "kind": "port",
            "host_cpus": ncpu,
            "sample": f"{n_items} items x 2ch x 10s@44.1kHz, stft(2048/512)+mel80+LUFS(IIR), best single pass "
                      f"({best:.2f} s) over thread counts <= {ncpu}; torch CPU ops + oracle/c DF-I lfilter (OpenMP)"}


def measured_traffic(n_local):
    """HBM bytes per launch of the fused STFT+mel kernel from the committed rocprofv3 PMC pass
    (profiles/r01_bench_pmc_summary.json; FETCH_SIZE and WRITE_SIZE collected in separate passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the
    configuration that was profiled (512 items on one GPU)."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_summary.json")
    if n_local != 512 or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        f = [v["FETCH_SIZE"] for k, v in d["pmc_fetch"].items() if "stft_mel_kernel" in k][0]
        w = [v["WRITE_SIZE"] for k, v in d["pmc_write"].items() if "stft_mel_kernel" in k][0]
        return (2.0 * f + w) * 1024.0
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="GLOBAL batch (items)")
    ap.add_argument("--cpu-items", type=int, default=32)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import audiotools_amd as A
    from audiotools_amd import _native, dist as adist


    rank, world, device = adist.init()
    if int(os.environ.get("LOCAL_RANK", 0)) == 0:
        _native.build()   # no-op when the in-tree library is up to date (it normally travels prebuilt)
    adist.barrier()
    assert device.type == "cuda", "bench.py needs a GPU (the product has no CPU fallback)"
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    lo, hi = adist.shard_range(args.batch, rank, world)
    n_local = hi - lo
    x = make_batch(n_local, device, 1234 + rank)
    sig = A.AudioSignal(x, SR)
    n_fft, hop = sig.stft_params.window_length, sig.stft_params.hop_length
    adist.broadcast_stft_mel_tables(SR, n_fft, "hann", N_MELS, device)

    T = x.shape[-1]
    rows = n_local * CH
    n_frames = 1 + T // hop
    F = n_fft // 2 + 1
    stft_bytes = rows * T * 4 + rows * n_frames * F * 8 + rows * n_frames * N_MELS * 4
    lufs_bytes = rows * T * 4 + n_local * 4

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        mel = sig.mel_spectrogram(N_MELS)
        if ev is not None:
            ev[1].record()
        sig._loudness = None
        lufs = sig.loudness()
        if ev is not None:
            ev[2].record()
        return mel, lufs

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    adist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        mel, lufs = step(events[k])
    torch.cuda.synchronize()
    adist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed[0])
    stft_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    lufs_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps
    assert torch.isfinite(mel).all() and torch.isfinite(lufs).all()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.batch * DUR / (elapsed / args.steps)
        achieved = stft_bytes / (stft_ms * 1e-3) / 1e9
        out = {
            "metric": "audio-seconds/sec (STFT+mel+LUFS pipeline), batch 512x2chx10s@44.1kHz",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"north-star: batch={args.batch} 2ch 10s@44.1kHz "
                                   f"mel_spectrogram(80) [fused STFT {n_fft}/{hop} hann + mel] + loudness()",
                       "global_batch": args.batch, "items_per_gpu": n_local, "parallelism": f"batch-shard x{world}",
                       "inputs": "device-resident (H2D excluded)"},
            "roofline": {"bound": "hbm", "kernel": "stft_mel_kernel<1024,4,true,4>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n_local),
                         "algorithmic_bytes_per_launch": stft_bytes, "avg_launch_ms": stft_ms,
                         "frac_of_measured_copy_6290": achieved / 6290.0},
            "kernels_ms": {"stft_mel": stft_ms, "lufs_total": lufs_ms,
                           "lufs_GBps": lufs_bytes / (lufs_ms * 1e-3) / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_items, args.cpu_iters)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
