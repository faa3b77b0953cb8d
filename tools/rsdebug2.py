#!/usr/bin/env python
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AT_RESAMPLE_F16_TUNE", "1")
import numpy as np
import torch
from audiotools_amd import _native, tables
from tests import synth

lib = _native.lib()
old, new = 441, 160
W, lo, _, _, width, NPB, NC, wk = tables.resample_f16_bank(old, new)
Wd, lod = torch.from_numpy(W.view(np.int32)).cuda(), torch.from_numpy(lo).cuda()


def run(xd, rp):
    os.environ["AT_RESAMPLE_F16_RP"] = rp
    rows, T = xd.shape[0] * xd.shape[1], xd.shape[-1]
    out_len = new * T // old
    y = torch.full((xd.shape[0], xd.shape[1], out_len), float("nan"), device="cuda")
    rc = lib.at_resample_f16s_f32(_native.ptr(xd), rows, T, _native.ptr(Wd), _native.ptr(lod), old, new, width, NPB, NC,
                                  int(lo.max()), wk, _native.ptr(y), out_len, _native.current_stream(xd.device))
    torch.cuda.synchronize()
    return y


def badtiles(y):
    b = ~torch.isfinite(y.reshape(-1, y.shape[-1]))
    out = []
    for r in range(b.shape[0]):
        idx = b[r].nonzero().flatten()
        if idx.numel():
            out.append((r, sorted(set((idx // 2560).tolist()))))
    return out


xs = synth.structured_batch(30011, 44100).cuda().contiguous()
for i in range(3):
    print("structured run", i, badtiles(run(xs, "1")))
ref = run(xs, "0")
y = run(xs, "1")
fin = torch.isfinite(y)
print("finite part equal to dma form:", bool((y[fin] == ref[fin]).all()))
# row 5 alone, at the same address alignment (3 words off)
big = torch.zeros(30011 + 8, device="cuda")
for off in (0, 1, 2, 3):
    v = big[off: off + 30011].view(1, 1, -1)
    v.copy_(xs[5])
    print("row 5 alone, offset", off, badtiles(run(v, "1")))
# every row alone
for r in range(6):
    print("row", r, "alone", badtiles(run(xs[r:r + 1].contiguous(), "1")))
# rows in reverse order
print("reversed", badtiles(run(xs.flip(0).contiguous(), "1")))
# a larger batch of the same rows: several tiles per workgroup
xl = xs.repeat(60, 1, 1).contiguous()
print("x60", badtiles(run(xl, "1"))[:12])
xr = (0.1 * torch.randn(64, 1, 30011, device="cuda"))
yr, y0 = run(xr, "1"), run(xr, "0")
print("random 64 rows: bad", badtiles(yr)[:8], "equal", bool(torch.equal(yr, y0)))
