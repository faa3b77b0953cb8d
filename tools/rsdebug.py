#!/usr/bin/env python
"""Development aid: where do the two forms of the fp16-split resampler differ from the oracle / each other?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AT_RESAMPLE_F16_TUNE", "1")
import numpy as np
import torch
from audiotools_amd import _native, tables
from oracle import restate
from tests import synth

lib = _native.lib()
old, new = 441, 160
W, lo, _, _, width, NPB, NC, wk = tables.resample_f16_bank(old, new)
Wd, lod = torch.from_numpy(W.view(np.int32)).cuda(), torch.from_numpy(lo).cuda()


def run(x, rp):
    os.environ["AT_RESAMPLE_F16_RP"] = rp
    rows, T = x.shape[0] * x.shape[1], x.shape[-1]
    out_len = new * T // old
    xd = x.cuda().contiguous()
    y = torch.full((x.shape[0], x.shape[1], out_len), float("nan"), device="cuda")
    rc = lib.at_resample_f16s_f32(_native.ptr(xd), rows, T, _native.ptr(Wd), _native.ptr(lod), old, new, width, NPB, NC,
                                  int(lo.max()), wk, _native.ptr(y), out_len, _native.current_stream(xd.device))
    torch.cuda.synchronize()
    return rc, y.cpu()


for name, x in (("structured", synth.structured_batch(30011, 44100)), ("T16", synth.audio_batch(3, 2, 16, seed=617, gaps=False, sample_rate=441)),
                ("ramp", (torch.arange(30011) / 30011.0 * 0.8).reshape(1, 1, -1)), ("const", torch.full((1, 1, 30011), 0.5)),
                ("alt", (0.25 + 0.25 * (1 - 2.0 * (torch.arange(30011) % 2))).reshape(1, 1, -1).float())):
    ref = restate.resample(x.double(), old, new)
    for rp in ("1", "0"):
        rc, y = run(x, rp)
        bad = ~torch.isfinite(y)
        print(f"{name} rp={rp} rc={rc} shape={tuple(y.shape)} non-finite={int(bad.sum())}", end="  ")
        for r in range(y.shape[0] * y.shape[1]):
            yy, rr = y.reshape(-1, y.shape[-1])[r], ref.reshape(-1, y.shape[-1])[r]
            b = (~torch.isfinite(yy)).nonzero().flatten()
            e = float((yy.double() - rr).abs()[torch.isfinite(yy)].max() / rr.abs().max()) if torch.isfinite(yy).any() else float("nan")
            print(f"[row {r}: err {e:.1e}" + (f", bad {b.numel()} in [{int(b.min())}, {int(b.max())}]" if b.numel() else "") + "]", end=" ")
        print()
    if name == "T16":
        print(" ref ", ref.flatten()[:10].tolist())
        print(" got ", y.flatten()[:10].tolist())
