#!/usr/bin/env python
"""Timing of the non-headline BASELINE.json configurations on one GPU (development aid).
cfg4: batch=1024 mono 5s@48kHz  Compose(LowPass, Equalizer, RoomImpulseResponse(2 s RIR))
cfg5: batch=2048 2ch 30s@44.1kHz resample 44.1k->16k + STFT + mel  (per-GPU share = 256 items)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiotools_amd as A
from audiotools_amd import transforms as tfm

ap = argparse.ArgumentParser()
ap.add_argument("--b4", type=int, default=1024)
ap.add_argument("--b5", type=int, default=256)
ap.add_argument("--only", default="lowpass,eq,applyir,chain,cfg5", help="comma list: lowpass,eq,applyir,chain,cfg5")
args = ap.parse_args()
only = set(args.only.split(","))
dev = "cuda"


def timed(fn, label, n=3):
    import gc
    fn(); torch.cuda.synchronize()
    gc.collect()        # a generation-2 collection of the thousands of parameter objects inside the timed loop read as
    gc.disable()        # "low_pass 20 ms, host enqueue 19.9 ms" in two sessions (s10, s18) and 0.75 ms in three others
    try:
        return _timed(fn, label, n)
    finally:
        gc.enable()


def _timed(fn, label, n):
    st0 = torch.cuda.memory_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host_ms = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    st1 = torch.cuda.memory_stats()
    dalloc = st1.get("num_device_alloc", 0) - st0.get("num_device_alloc", 0)
    dfree = st1.get("num_device_free", 0) - st0.get("num_device_free", 0)
    print(f"{label:34s} {ms:9.2f} ms   (host enqueue {host_ms:7.2f} ms; gpu span {e0.elapsed_time(e1) / n:8.2f} ms; hipMalloc {dalloc}, hipFree {dfree}, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB)", flush=True)
    return ms


# ---- cfg4
B, T, SR = args.b4, 240000, 48000
g = torch.Generator(device=dev).manual_seed(1)
x = (0.1 * torch.randn(B, 1, T, device=dev, generator=g)).clamp_(-1, 1)
bank = torch.randn(64, 1, 96000, device=dev, generator=g) * torch.exp(-torch.arange(96000, device=dev) / (0.3 * SR))
chain = tfm.Compose(tfm.LowPass(cutoff=("choice", [4000, 8000, 16000])), tfm.Equalizer(n_bands=6),
                    tfm.RoomImpulseResponse(loader=tfm.TensorLoader(bank, SR), duration=2.0))
sig = A.AudioSignal(x, SR)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kw = chain.batch_instantiate(list(range(B)), A.AudioSignal(x[:1], SR))
    t_host = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print(f"cfg4 batch_instantiate #{rep}: host {t_host:7.1f} ms, with the device work {(time.perf_counter() - t0) * 1e3:7.1f} ms", flush=True)
kw = A.util.prepare_batch(kw, dev)
kc = kw["Compose"]
# (nothing below writes into x: every stage produces a new tensor, as in bench.py's cfg4 step)
if "lowpass" in only:
    if os.environ.get("CFGBENCH_PROFILE"):
        import cProfile, pstats
        A.AudioSignal(x, SR).low_pass(kc["0.LowPass"]["cutoff"]); torch.cuda.synchronize()
        pr = cProfile.Profile(); pr.enable()
        A.AudioSignal(x, SR).low_pass(kc["0.LowPass"]["cutoff"]); torch.cuda.synchronize()
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    timed(lambda: A.AudioSignal(x, SR).low_pass(kc["0.LowPass"]["cutoff"]), "cfg4 low_pass (B per-item cutoffs)")
if "eq" in only:
    timed(lambda: A.AudioSignal(x, SR).equalizer(kc["1.Equalizer"]["eq"]), "cfg4 equalizer (6 bands)")
ir = kc["2.RoomImpulseResponse"]["ir_signal"]
if "applyir" in only:
    rir = chain.transforms[2]
    kw_rir = {rir.name: kc[rir.name]}
    timed(lambda: rir(A.AudioSignal(x, SR), **kw_rir), "cfg4 RoomImpulseResponse transform (apply_ir, 2 s RIR, DRR, EQ)")
    timed(lambda: A.AudioSignal(x, SR).apply_ir(ir.clone(), kc["2.RoomImpulseResponse"]["drr"], kc["2.RoomImpulseResponse"]["eq"]),
          "cfg4 apply_ir called directly (+ the caller's ir.clone(), + padding of the IR object)")
if "chain" in only:
    ms = timed(lambda: chain(A.AudioSignal(x, SR), **kw), "cfg4 full chain")
    print(f"cfg4 throughput: {B * 5.0 / (ms * 1e-3):.0f} audio-seconds/sec")
if "cfg5" not in only:
    sys.exit(0)

# ---- cfg5 (per-GPU share)
del x, sig, kw, kc, ir
torch.cuda.empty_cache()
B, T, SR = args.b5, 1323000, 44100
x = (0.1 * torch.randn(B, 2, T, device=dev, generator=g)).clamp_(-1, 1)


def cfg5():
    s = A.AudioSignal(x, SR).resample(16000)
    return s.mel_spectrogram(80)


timed(lambda: A.AudioSignal(x, SR).resample(16000), "cfg5 resample 44.1k->16k")
ms = timed(cfg5, "cfg5 resample + stft + mel")
print(f"cfg5 throughput: {B * 30.0 / (ms * 1e-3):.0f} audio-seconds/sec per GPU")
