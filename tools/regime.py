#!/usr/bin/env python
"""What decides the speed of the north-star STFT + mel kernel's traffic on a given box / in a given process?
(development aid, round 5.)  The kernel's zero-compute twin (at_stft_mel_floor_f32: same grid, schedule, addresses, load /
store instructions, no arithmetic) was seen in two regimes, 1.65-1.73 ms and 2.0-2.1 ms, between boxes and between
processes on one box.  This tool separates the candidate causes inside ONE process:

  placement   N independent buffer sets (each a fresh allocation while the earlier ones stay alive, so the physical pages
              differ), the twin + the kernel + a plain copy on each, then set 0 again (drift check); one set from raw
              hipMalloc instead of torch's caching allocator
  schedule    (development build, AT_DEV_KNOBS=1) run length / XCD spans / store policy of the twin on set 0
  clocks      rocm-smi / amd-smi samples while a long queue of launches runs

usage: AT_DEV_KNOBS=1 python tools/regime.py [--sets 4] [--iters 12] [--smi] [--sched] [--tag X]
Under `rocprofv3 --pmc ...` use --pmc: few launches per set, one label line per set so that the counter rows can be
matched to sets by launch order."""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiotools_amd import _native, tables

ap = argparse.ArgumentParser()
ap.add_argument("--sets", type=int, default=4)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--smi", action="store_true")
ap.add_argument("--sched", action="store_true")
ap.add_argument("--pmc", action="store_true")
ap.add_argument("--rawmalloc", action="store_true")
ap.add_argument("--combo", action="store_true", help="mix x / spectrum / mel buffers of the slowest and the fastest set")
ap.add_argument("--slices", action="store_true", help="the twin on eighths of the rows of the slowest and the fastest set")
ap.add_argument("--variants", action="store_true", help="(dev build) twin without loads / without stores on both")
ap.add_argument("--pool", type=int, default=0, help="carve N extra sets out of ONE hipMalloc of N x the set size")
ap.add_argument("--spans", action="store_true", help="(dev build) span schedules: phase-shifted rounds, row-interleaved spans")
ap.add_argument("--affinity", action="store_true", help="(dev build) one XCD at a time on each eighth of the spectrum buffer")
ap.add_argument("--spacing", action="store_true", help="the twin on the first R rows of every set, R = 1024 ... 512: the spacing of the eight XCD spans changes with R")
ap.add_argument("--tag", default="")
args = ap.parse_args()

dev = torch.device("cuda")
B, C, SR, n_fft, hop, n_mels = args.batch, 2, 44100, 2048, 512, 80
T = 10 * SR
rows, N, F = B * C, 1 + T // hop, n_fft // 2 + 1
nx, ns, nm = rows * T * 4, rows * N * F * 8, rows * N * n_mels * 4
ALG = nx + ns + nm
win = tables.window("hann", n_fft, dev)
tw = tables.stft_twiddles(n_fft, dev)
info, w = tables.mel_units(SR, n_fft, n_mels, 0.0, None, dev)
lib = _native.lib()
_native.dev_lib()          # binds the measurement entry points (the tools run the development build: same handle)
st = _native.current_stream(dev)
sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()[:16]
print(f"# regime {args.tag} lib={os.path.basename(_native.LIB_PATH)} sha256={sha} B={B} alg_bytes={ALG} dev={torch.cuda.get_device_name(0)}", flush=True)


def launch(fn, px, ps, pm, rows=rows):
    rc = fn(ctypes.c_void_p(px), rows, T, _native.ptr(win), _native.ptr(tw), n_fft, hop, 0, 0, 1, 0, N, ctypes.c_void_p(ps),
            _native.ptr(info), _native.ptr(w), int(info.shape[0]), n_mels, ctypes.c_void_p(pm), st)
    assert rc == 0, rc


def timeit(fn, iters=None, warm=2):
    iters = iters or args.iters
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


xsrc = (0.1 * torch.randn(rows, T, device=dev)).clamp_(-1, 1)


class BufSet:
    def __init__(self, how="torch", ptrs=None):
        self.how = how
        if ptrs is not None:
            self.px, self.ps, self.pm = ptrs
            hip = ctypes.CDLL("libamdhip64.so")
            rc = hip.hipMemcpy(ctypes.c_void_p(self.px), ctypes.c_void_p(xsrc.data_ptr()), ctypes.c_size_t(nx), 3)
            assert rc == 0, rc
        elif how == "torch":
            self.x = torch.empty(rows, T, device=dev)
            self.s = torch.empty(ns // 4, device=dev)
            self.m = torch.empty(nm // 4, device=dev)
            self.px, self.ps, self.pm = self.x.data_ptr(), self.s.data_ptr(), self.m.data_ptr()
            self.x.copy_(xsrc)
        else:
            hip = ctypes.CDLL("libamdhip64.so")
            ptrs = []
            for nb in (nx, ns, nm):
                p = ctypes.c_void_p()
                rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nb))
                assert rc == 0, rc
                ptrs.append(p.value)
            self.px, self.ps, self.pm = ptrs
            rc = hip.hipMemcpy(ctypes.c_void_p(self.px), ctypes.c_void_p(xsrc.data_ptr()), ctypes.c_size_t(nx), 3)
            assert rc == 0, rc

    def twin(self):
        launch(lib.at_stft_mel_floor_f32, self.px, self.ps, self.pm)

    def kern(self):
        launch(lib.at_stft_mel_f32, self.px, self.ps, self.pm)


def line(label, bs, extra=""):
    t_twin = timeit(bs.twin)
    t_kern = timeit(bs.kern)
    t_twin2 = timeit(bs.twin)
    msg = (f"{label:28s} x={bs.px:#x} s={bs.ps:#x} m={bs.pm:#x}  twin {t_twin:.3f} / {t_twin2:.3f} ms  kernel {t_kern:.3f} ms"
           f"  ({100 * ALG / t_kern / 1e6 / 8000:.1f} % of 8 TB/s; twin {100 * ALG / min(t_twin, t_twin2) / 1e6 / 8000:.1f} %)")
    if bs.how == "torch":
        t_copy = timeit(lambda: bs.s[: ns // 8].copy_(bs.s[ns // 8:]), iters=6)
        t_fill = timeit(lambda: bs.s.fill_(0.5), iters=6)
        msg += f"  copy_ {ns / t_copy / 1e9:.2f} TB/s  fill_ {ns / t_fill / 1e9:.2f} TB/s"
    print(msg + extra, flush=True)
    return t_twin, t_kern


sets = []
if args.pmc:
    # few launches, recognisable order: per set 3 x twin then 3 x kernel
    for i in range(args.sets):
        bs = BufSet()
        sets.append(bs)
        for _ in range(3):
            bs.twin()
        for _ in range(3):
            bs.kern()
        torch.cuda.synchronize()
        print(f"pmc set {i}: x={bs.px:#x} s={bs.ps:#x} m={bs.pm:#x} (3 twin launches then 3 kernel launches)", flush=True)
    sys.exit(0)

for i in range(args.sets):
    bs = BufSet()
    sets.append(bs)
    line(f"set {i} (torch.empty)", bs)
line("set 0 again", sets[0])
if args.rawmalloc:
    raw = BufSet("hip")
    line("raw hipMalloc", raw)
    line("set 1 again", sets[min(1, len(sets) - 1)])

if args.pool:
    hip = ctypes.CDLL("libamdhip64.so")
    per = ((nx + (1 << 21) - 1) >> 21 << 21) + ((ns + (1 << 21) - 1) >> 21 << 21) + ((nm + (1 << 21) - 1) >> 21 << 21)
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(per * args.pool + (1 << 21)))
    assert rc == 0, rc
    base = (p.value + (1 << 21) - 1) >> 21 << 21
    for i in range(args.pool):
        b0 = base + i * per
        bs = BufSet("pool", (b0, b0 + ((nx + (1 << 21) - 1) >> 21 << 21), b0 + per - ((nm + (1 << 21) - 1) >> 21 << 21)))
        sets.append(bs)
        line(f"pool set {i} (one hipMalloc)", bs)

times = []
for i, bs in enumerate(sets):
    times.append((timeit(bs.twin), i))
slow, fast = sets[max(times)[1]], sets[min(times)[1]]
print(f"slowest set {max(times)[1]} ({max(times)[0]:.3f} ms), fastest set {min(times)[1]} ({min(times)[0]:.3f} ms)", flush=True)

if args.combo:
    for lx, bx in (("slow", slow), ("fast", fast)):
        for ls, b_s in (("slow", slow), ("fast", fast)):
            for lm, bm in (("slow", slow), ("fast", fast)):
                t_twin = timeit(lambda: launch(lib.at_stft_mel_floor_f32, bx.px, b_s.ps, bm.pm))
                t_kern = timeit(lambda: launch(lib.at_stft_mel_f32, bx.px, b_s.ps, bm.pm))
                print(f"combo x={lx} spectrum={ls} mel={lm}: twin {t_twin:.3f} ms  kernel {t_kern:.3f} ms", flush=True)

if args.slices:
    NSL = 8
    rs = rows // NSL
    for lab, bs in (("slow", slow), ("fast", fast)):
        parts = []
        for k in range(NSL):
            px, ps, pm = bs.px + k * rs * T * 4, bs.ps + k * rs * N * F * 8, bs.pm + k * rs * N * n_mels * 4
            parts.append(timeit(lambda: launch(lib.at_stft_mel_floor_f32, px, ps, pm, rows=rs), iters=30, warm=5))
        print(f"slices of the {lab} set (twin on {rs} rows each, ms): " + " ".join(f"{v:.4f}" for v in parts) + f"  sum {sum(parts):.3f}", flush=True)

if args.variants:
    assert os.environ.get("AT_STFT_TUNE") == "1", "start the process with AT_STFT_TUNE=1"
    for lab, bs in (("slow", slow), ("fast", fast)):
        for name, fl in (("twin nt (shipped)", 1), ("twin plain stores", 0), ("twin nt, no loads", 65), ("twin plain, no loads", 64),
                         ("twin no stores (loads only)", 129)):
            os.environ["AT_STFT_FLAGS"] = str(fl)
            t1 = timeit(bs.twin)
            print(f"variant on the {lab} set: {name:28s} {t1:.3f} ms", flush=True)
    os.environ.pop("AT_STFT_FLAGS", None)

if args.spacing:
    print("spacing sweep: twin on the first R rows of each set (8 spans of R / 8 rows = R x 0.8836 MB apart); us per row", flush=True)
    Rs = [1024, 1016, 1008, 1000, 992, 976, 960, 944, 928, 912, 896, 864, 832, 800, 768, 704, 640, 576, 512]
    print("      R: " + " ".join(f"{r:6d}" for r in Rs), flush=True)
    for i, bs in enumerate(sets):
        vals = []
        for R in Rs:
            t_ = timeit(lambda: launch(lib.at_stft_mel_floor_f32, bs.px, bs.ps, bs.pm, rows=R), iters=8)
            vals.append(1e3 * t_ / R)
        print(f"  set {i}: " + " ".join(f"{v:6.3f}" for v in vals), flush=True)

if args.spacing:
    for Rw in (512, 256):
        offs = list(range(0, rows - Rw + 1, 64))
        print(f"window sweep: twin on rows [a, a + {Rw}) of each set; us per row;  a = " + " ".join(f"{a:5d}" for a in offs), flush=True)
        for i, bs in enumerate(sets):
            vals = []
            for a in offs:
                px, ps, pm = bs.px + a * T * 4, bs.ps + a * N * F * 8, bs.pm + a * N * n_mels * 4
                t_ = timeit(lambda: launch(lib.at_stft_mel_floor_f32, px, ps, pm, rows=Rw), iters=8)
                vals.append(1e3 * t_ / Rw)
            print(f"  set {i}:                                                       " + " ".join(f"{v:5.3f}" for v in vals), flush=True)

if args.affinity:
    assert os.environ.get("AT_STFT_TUNE") == "1", "start the process with AT_STFT_TUNE=1"
    NSL = 8
    rs = rows // NSL
    os.environ["AT_STFT_NX"] = "1"
    for lab, bs in (("slow", slow), ("fast", fast)):
        print(f"affinity on the {lab} set: rows = XCD 0..7 (alone), columns = eighth of the buffer it writes; ms per launch (twin, nt stores)", flush=True)
        for k in range(8):
            os.environ["AT_STFT_FLAGS"] = str(1 | ((k + 1) << 10))
            parts = []
            for j in range(NSL):
                px, ps, pm = bs.px + j * rs * T * 4, bs.ps + j * rs * N * F * 8, bs.pm + j * rs * N * n_mels * 4
                parts.append(timeit(lambda: launch(lib.at_stft_mel_floor_f32, px, ps, pm, rows=rs), iters=20, warm=3))
            print(f"  xcd {k}: " + " ".join(f"{v:.4f}" for v in parts), flush=True)
        os.environ["AT_STFT_FLAGS"] = "1"
        parts = []
        for j in range(NSL):
            px, ps, pm = bs.px + j * rs * T * 4, bs.ps + j * rs * N * F * 8, bs.pm + j * rs * N * n_mels * 4
            parts.append(timeit(lambda: launch(lib.at_stft_mel_floor_f32, px, ps, pm, rows=rs), iters=20, warm=3))
        print("  all 8: " + " ".join(f"{v:.4f}" for v in parts), flush=True)
    for k in ("AT_STFT_NX", "AT_STFT_FLAGS"):
        os.environ.pop(k, None)

if args.spans:
    assert os.environ.get("AT_STFT_TUNE") == "1", "start the process with AT_STFT_TUNE=1"
    os.environ["AT_STFT_NX"] = "0"
    for i, bs in enumerate(sets):
        res = []
        for mul in range(4):
            for xr in range(8):
                os.environ["AT_STFT_FLAGS"] = str(1 | (xr << 14) | (mul << 17))
                res.append(timeit(bs.twin, iters=6))
        print(f"span permutations, set {i}: twin ms for span = (xcd * m) ^ x, rows m = 1, 3, 5, 7, columns x = 0..7", flush=True)
        for mul in range(4):
            print("   " + " ".join(f"{v:.3f}" for v in res[8 * mul: 8 * mul + 8]), flush=True)
    for rep in range(1):
        for i, bs in enumerate(sets):
            for name, fl, nxv in (("shipped", 1, 0), ("phase-shifted rounds", 257, 0), ("row-interleaved spans", 513, 0), ("phase-shifted, 16 spans", 257, 16),
                                  ("phase-shifted, 4 spans", 257, 4)):
                os.environ["AT_STFT_FLAGS"] = str(fl)
                os.environ["AT_STFT_NX"] = str(nxv)
                t_twin = timeit(bs.twin, iters=8)
                t_kern = timeit(bs.kern, iters=8)
                print(f"spans rep {rep} set {i}: {name:26s} twin {t_twin:.3f} ms  kernel {t_kern:.3f} ms", flush=True)
    for k in ("AT_STFT_NX", "AT_STFT_FLAGS"):
        os.environ.pop(k, None)

if args.sched:
    assert os.environ.get("AT_STFT_TUNE") == "1", "start the process with AT_STFT_TUNE=1 (the library reads the switch once, at its first call)"
    for rep in range(2):
        for lab, bs in (("slow", slow), ("fast", fast)):
            for runmax, nxv in [(72, 0), (72, 1), (72, 2), (72, 4), (72, 16), (72, 32), (16, 0), (16, 1), (8, 1), (4, 1), (2, 1), (431, 0), (431, 1), (144, 1)]:
                os.environ["AT_STFT_RUNMAX"] = str(runmax)
                os.environ["AT_STFT_NX"] = str(nxv)
                t_twin = timeit(bs.twin, iters=8)
                t_kern = timeit(bs.kern, iters=8)
                print(f"sched rep {rep} {lab} set: runmax {runmax:4d} nx {nxv:2d}:  twin {t_twin:.3f} ms  kernel {t_kern:.3f} ms", flush=True)
    for k in ("AT_STFT_RUNMAX", "AT_STFT_NX", "AT_STFT_FLAGS"):
        os.environ.pop(k, None)

if args.smi:
    def smi():
        out = {}
        for cmd in (["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], ["amd-smi", "metric", "-c", "-p", "--json"]):
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                out[cmd[0]] = r.stdout.strip()[:3000] if r.returncode == 0 else f"rc={r.returncode} {r.stderr[:200]}"
            except Exception as e:          # tool missing in the image
                out[cmd[0]] = repr(e)
        return out

    print("smi idle:", json.dumps(smi()), flush=True)
    for name, fn in (("twin", sets[0].twin), ("kernel", sets[0].kern)):
        t0 = time.time()
        for _ in range(1500):          # ~3 s of queued launches
            fn()
        s1 = smi()
        s2 = smi()
        torch.cuda.synchronize()
        print(f"smi during {name} (queue of 1500 launches drained in {time.time() - t0:.2f} s):", json.dumps(s1), flush=True)
        print(f"smi during {name} (2):", json.dumps(s2), flush=True)
    line("set 0 after the long queues", sets[0])
