"""Host-side logic of the drop-in API on CPU tensors (torch path): container semantics,
table design against the oracle leaves, cache invalidation rules."""
import os
import numpy as np
import pytest
import torch

import audiotools_amd as A
from audiotools_amd import tables
from oracle import restate
from oracle.leaves import julius_leaf, misc_leaves, pyloudnorm_leaf
from tests import synth


def test_tables_match_leaves():
    for sr, n_fft, nm in [(44100, 2048, 80), (16000, 512, 40), (48000, 1024, 128)]:
        assert np.array_equal(tables.mel_filters_np(sr, n_fft, nm), misc_leaves.librosa_mel(sr=sr, n_fft=n_fft, n_mels=nm))
    assert np.array_equal(tables.mel_filters_np(22050, 1024, 64, 50.0, 8000.0),
                          misc_leaves.librosa_mel(sr=22050, n_fft=1024, n_mels=64, fmin=50.0, fmax=8000.0))
    for fc in ("K-weighting", "Fenton/Lee 1", "Dash et al."):
        sos, g = tables.weighting_sos(48000, fc)
        ref = pyloudnorm_leaf.Meter(48000, fc)._filters
        for row, f in zip(sos, ref.values()):
            assert np.allclose(row[:3], f.b, rtol=0, atol=1e-15) and np.allclose(row[3:], f.a, rtol=0, atol=1e-15)
    # ITU-R BS.1770 48 kHz table (4-digit agreement, the known pyloudnorm-vs-ITU gap)
    sos, _ = tables.weighting_sos(48000)
    assert np.allclose(sos[0], [1.53512486, -2.69169619, 1.19839281, 1.0, -1.69065929, 0.73248077], atol=2e-4)
    assert np.allclose(sos[1][3:], [1.0, -1.99004745, 0.99007225], atol=2e-4)
    for c in (4000 / 48000, 0.25, 50 / 44100):
        ct = torch.tensor(c, dtype=torch.float32)
        ours = tables.lowpass_taps(ct, 51)
        ref = julius_leaf.LowPassFilters([ct], zeros=51).filters[0, 0]
        assert torch.equal(ours, ref)
    bank, old, new, width = tables.resample_bank(44100, 16000)
    assert (old, new, width) == (441, 160, 70) and bank.shape == (160, 581)
    assert torch.equal(bank, julius_leaf.ResampleFrac(44100, 16000).kernel[:, 0])
    assert np.allclose(tables.dct_np(40, 80), misc_leaves.create_dct(40, 80, "ortho").numpy(), atol=2e-7)
    assert np.array_equal(tables.window_np("sqrt_hann", 512), restate.get_window("sqrt_hann", 512).numpy())


def test_mel_units_reconstruct_basis():
    from audiotools_amd import _native

    _native.build()
    for sr, n_fft, nm in [(44100, 2048, 80), (16000, 512, 80), (8000, 256, 40), (44100, 64, 20)]:
        basis = tables.mel_filters_np(sr, n_fft, nm)
        info, w = tables.mel_units_np(basis)
        assert info.shape[0] in (128, 256, 384)
        rec = np.zeros_like(basis)
        stored = set()
        for u in range(info.shape[0]):
            row, m, flags = info[u, 0] & 0xffff, (info[u, 0] >> 16) & 0xffff, info[u, 1]
            if m == 0xffff:
                assert flags == 0 and not w[u].any()
                continue
            hi = min(16 * row + 16, basis.shape[1])
            rec[m, 16 * row:hi] += w[u][: hi - 16 * row]
            if flags & 16:
                assert m not in stored
                stored.add(m)
        assert np.array_equal(rec, basis)
        assert stored == set(range(nm))
        # emulate the in-wave shuffle-down tree: every band's units end up summed in its first lane
        vals = np.arange(1, info.shape[0] + 1, dtype=np.float64)
        acc = vals.copy()
        for step, d in enumerate((1, 2, 4, 8)):
            sh = np.concatenate([acc[d:], np.zeros(d)])
            sh[(np.arange(len(sh)) % 16) + d >= 16] = 0.0   # DPP row_shl stays inside a 16-lane row
            take = (info[:, 1] >> step) & 1
            acc = acc + np.where(take == 1, sh, 0.0)
        for m in range(nm):
            us = [u for u in range(info.shape[0]) if ((info[u, 0] >> 16) & 0xffff) == m]
            assert us == list(range(us[0], us[-1] + 1)) and us[0] // 16 == us[-1] // 16
            assert acc[us[0]] == vals[us].sum()


def test_mel_unit_reads_are_bank_conflict_free():
    """The unit rounds of the fused mel stage read 16 magnitudes per lane as four ds_read_b128 from rows of 20 floats.  With
    gfx950's b128 lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS) the
    north-star table (44.1 kHz, n_fft 2048, 80 bands) must cost 4 LDS cycles per read instruction and round -- padding units
    read their predecessor's row (a broadcast); pointing at row 0 they collided with rows 16 / 32 / 48 (84 cycles per frame
    instead of 64, round 6)."""
    from audiotools_amd import _native

    _native.build()
    info, _w = tables.mel_units_np(tables.mel_filters_np(44100, 2048, 80))
    rows = info[:, 0] & 0xffff
    g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    groups = [g0, g1, [l + 32 for l in g0], [l + 32 for l in g1]]
    total = 0
    for r in range(info.shape[0] // 64):
        for i4 in range(4):
            for g in groups:
                banks = {}
                for lane in g:
                    a0 = int(rows[r * 64 + lane]) * 20 + 4 * i4
                    for d in range(4):
                        banks.setdefault((a0 + d) % 64, set()).add(a0 + d)
                total += max(len(v) for v in banks.values())
    assert total == 4 * 4 * (info.shape[0] // 64), total


def test_mel_bands_reconstruct_basis():
    """Banded form for the generic-size fused mel stage (at_mel_bands_host): 16-bin chunks starting at multiples of 4
    bins (16-byte LDS reads of the magnitudes), every band's chunks consecutive; the chunks rebuild the dense basis and
    never reach past bin n_bins - 1 + 15 (the zero slack of the kernel's |X| rows)."""
    from audiotools_amd import _native

    _native.build()
    for sr, n_fft, nm in [(96000, 4096, 80), (192000, 8192, 128), (16000, 400, 40), (48000, 1920, 64)]:
        basis = tables.mel_filters_np(sr, n_fft, nm)
        F = basis.shape[1]
        info, w = tables.mel_bands_np(basis)
        n = w.shape[0]
        first, bands = info[:n], info[n:].reshape(nm, 2)
        assert np.all(first % 4 == 0) and np.all(first >= 0) and np.all(first + 15 <= F - 1 + 15)
        rec = np.zeros((nm, F + 16), dtype=np.float32)
        nxt = 0
        for m in range(nm):
            c0, cn = bands[m]
            assert c0 == nxt
            nxt += cn
            for c in range(c0, c0 + cn):
                assert first[c] == first[c0] + 16 * (c - c0)
                rec[m, first[c]: first[c] + 16] += w[c]
        assert nxt == n
        assert np.array_equal(rec[:, :F], basis) and not rec[:, F:].any()


def test_cpu_path_matches_oracle():
    x = synth.audio_batch(3, 2, 22050, seed=3, gaps=False)
    s = A.AudioSignal(x.clone(), 44100)
    assert torch.equal(s.stft(), restate.stft(x, 2048, 512))
    assert torch.equal(s.mel_spectrogram(80), restate.mel_spectrogram(restate.stft(x, 2048, 512), 44100, 80))
    assert torch.allclose(s.loudness(), restate.loudness(x, 44100), atol=1e-4)
    assert torch.allclose(s.clone().istft().audio_data, x, atol=1e-6)
    cut = torch.tensor([4000.0, 8000.0, 1000.0])
    assert torch.allclose(s.clone().low_pass(cut).audio_data, restate.low_pass(x, cut, 44100), atol=1e-6)
    assert torch.allclose(s.clone().high_pass(cut).audio_data, restate.high_pass(x, cut, 44100), atol=1e-6)
    db = -torch.rand(3, 6)
    assert torch.allclose(s.clone().equalizer(db).audio_data, restate.equalizer(x, 44100, db), atol=1e-6)
    assert torch.equal(s.clone().resample(16000).audio_data, restate.resample(x, 44100, 16000))
    ir = torch.randn(3, 1, 4000) * torch.exp(-torch.arange(4000) / 500.0)
    got = s.clone().convolve(A.AudioSignal(ir.clone(), 44100)).audio_data
    assert torch.allclose(got, restate.convolve(x, ir), atol=1e-5)


def test_container_semantics():
    x = synth.audio_batch(4, 1, 16000, seed=4, gaps=False)
    s = A.AudioSignal(x.clone(), 16000)
    assert s.stft_params == A.STFTParams(512, 128, "hann", False, "reflect")
    s.stft()
    assert s.stft_data is not None
    s.loudness()
    assert s._loudness is not None
    s.audio_data = s.audio_data * 0.5            # setter drops the cached loudness
    assert s._loudness is None
    s.stft_params = A.STFTParams(256, 64)        # setter drops stft_data
    assert s.stft_data is None and s.stft_params.window_type == "hann"
    s.stft()
    s.low_pass(2000)                             # low_pass drops stft_data
    assert s.stft_data is None
    s.resample(8000)                             # resample keeps stft_params (bug-compatible)
    assert s.sample_rate == 8000 and s.stft_params.window_length == 256 and s.signal_length == 8000
    with pytest.raises(RuntimeError):
        A.AudioSignal(x.clone(), 16000).istft()
    with pytest.raises(AssertionError):
        A.AudioSignal(x.clone())
    with pytest.raises(ValueError):
        A.AudioSignal(3.0, 16000)
    with pytest.raises(AssertionError):
        A.AudioSignal(x.clone(), 16000).stft(512, 100, match_stride=True)
    sub = A.AudioSignal(x.clone(), 16000)[1:3]
    assert sub.batch_size == 2
    b = A.AudioSignal.batch([A.AudioSignal(x[i: i + 1, :, : 16000 - 100 * i].clone(), 16000) for i in range(3)],
                            pad_signals=True)
    assert b.shape == (3, 1, 16000)
    with pytest.raises(RuntimeError):
        A.AudioSignal.batch([A.AudioSignal(x[:1].clone(), 16000), A.AudioSignal(x[:1].clone(), 8000)])
    m = torch.tensor([True, False, True, False])
    t = A.AudioSignal(x.clone(), 16000)
    t[m] = t[m] * 0.0
    assert t.audio_data[0].abs().max() == 0 and t.audio_data[1].abs().max() > 0
    assert (A.AudioSignal(x.clone(), 16000) * 2 - A.AudioSignal(x.clone(), 16000)) == A.AudioSignal(x.clone(), 16000)


def test_tap_designs_on_host():
    """The batched tap designs that feed at_fir_fft_f32 / at_fir_per_item_f32 (run here on CPU
    tensors; the same code runs on the device) against the julius restatement of the oracle."""
    from audiotools_amd import fx, kernels
    from oracle.leaves import julius_leaf
    x = synth.audio_batch(3, 1, 6000, seed=71, gaps=False)
    # per-item windowed-sinc low-pass (dsp.py:177-179): common centred grid, per-item lengths
    cut = torch.tensor([4000.0, 900.0, 16000.0]) / 44100
    taps = kernels.sinc_taps_batched(cut, 51)
    L = taps.shape[-1]
    xp = torch.nn.functional.pad(x, ((L - 1) // 2, (L - 1) // 2), mode="replicate")
    got = torch.stack([torch.nn.functional.conv1d(xp[b][None], taps[b][None, None])[0] for b in range(3)])
    ref = torch.stack([julius_leaf.LowPassFilter(float(c), zeros=51)(x[b][None])[0] for b, c in enumerate(cut)])
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    with pytest.raises(ValueError):
        kernels.sinc_taps_batched(torch.tensor([0.6]), 51)
    with pytest.raises(ValueError):
        kernels.sinc_taps_batched(torch.tensor([-0.1]), 51)
    # equalizer collapsed to ONE composite FIR per item (effects.py:399-433)
    w = 10 ** (-torch.rand(3, 6, generator=torch.Generator().manual_seed(4)))
    ctaps, half = fx.equalizer_taps(44100, w)
    xp = torch.nn.functional.pad(x, (half, half), mode="replicate")
    got = torch.stack([torch.nn.functional.conv1d(xp[b][None], ctaps[b][None, None])[0] for b in range(3)])
    bands = fx.band_split_torch(x, 44100, 6)                 # (6, B, C, T), sums to x
    ref = (bands * w.T[:, :, None, None]).sum(0)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    assert kernels.istft_fused_supported(2048, 512) and kernels.istft_fused_supported(512, 32)
    assert not kernels.istft_fused_supported(512, 100) and not kernels.istft_fused_supported(512, 512)


def test_native_dispatch_rules_on_cpu():
    """CPU tensors never reach the C ABI: the object API runs the torch formulation, the launchers
    refuse (NativeError), and the spectral-edit / autograd dispatch predicates are False."""
    from audiotools_amd import _native, filters, kernels, spectral
    x = synth.audio_batch(2, 1, 4096, seed=72, gaps=False)
    s = A.AudioSignal(x.clone(), 16000, stft_params=A.STFTParams(512, 128))
    X = s.stft()
    assert not kernels.is_native(x) and not kernels.spec_native(X)
    assert not filters._per_item_native(X, 100.0, 200.0)
    assert not spectral._native_autograd_ok(x.clone().requires_grad_(True), 512, 128, False)
    for fn in (lambda: kernels.stft_mel(x, torch.hann_window(512), 512, 128), lambda: kernels.absmax(x),
               lambda: kernels.fir_per_item(x, torch.ones(1, 9)), lambda: kernels.integrated_loudness(x, 16000)):
        with pytest.raises(_native.NativeError):
            fn()
    # masks / phase shift on CPU = the reference's polar formulation; new tensor each time
    before = s.stft_data
    s.mask_frequencies(1000.0, 2000.0)
    assert s.stft_data is not before
    y = s.istft().audio_data
    assert y.shape == x.shape and torch.isfinite(y).all()


def test_device_stager_passthrough_and_order():
    """DeviceStager yields every batch once, in order (CPU device = pass-through; the CUDA path with
    pinned buffers and events is covered by the GPU suite)."""
    from audiotools_amd.data import DeviceStager
    batches = [torch.full((2, 1, 8), float(i)) for i in range(5)]
    got = [b.clone() for b in DeviceStager(batches, "cpu")]
    assert len(got) == 5 and all(torch.equal(g, b) for g, b in zip(got, batches))
    assert list(DeviceStager([], "cpu")) == []


def test_mixed_radix_butterflies_on_host(tmp_path):
    """Every radix butterfly of csrc/generic_fft.h (2, 3, 4, 5, 7 and the composite 8, 9, 16, 25) against an O(R^2)
    float64 DFT: the gfft namespace is lifted out of the header and compiled for the HOST with g++ (the device
    qualifiers defined away), so the constants and the index algebra are checked without a GPU."""
    import re
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "audiotools_amd", "csrc", "generic_fft.h")).read()
    body = src[src.index("namespace gfft {"): src.index("// ---- in-place mixed-radix passes over an LDS tile")] + "}\n"
    prog = r"""
#include <cmath>
#include <cstdio>
#include <complex>
#include <random>
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
#define __device__
#define __forceinline__ inline
""" + body + r"""
using namespace gfft;
template <int R> double check() {
  std::mt19937 g(R); std::normal_distribution<float> d;
  float2 v[MAX_RADIX]; std::complex<double> x[MAX_RADIX];
  for (int i = 0; i < R; ++i) { v[i] = make_float2(d(g), d(g)); x[i] = {v[i].x, v[i].y}; }
  dft_r<R>(v);
  double worst = 0;
  for (int k = 0; k < R; ++k) {
    std::complex<double> s = 0;
    for (int n = 0; n < R; ++n) s += x[n] * std::polar(1.0, -2.0 * M_PI * k * n / R);
    worst = std::fmax(worst, std::abs(s - std::complex<double>(v[k].x, v[k].y)));
  }
  return worst;
}
int main() {
  printf("%.3e %.3e %.3e %.3e %.3e %.3e %.3e %.3e %.3e\n", check<2>(), check<3>(), check<4>(), check<5>(), check<7>(),
         check<8>(), check<9>(), check<16>(), check<25>());
}
"""
    cpp = tmp_path / "dft.cpp"
    cpp.write_text(prog)
    exe = tmp_path / "dft"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(cpp)])
    errs = [float(v) for v in subprocess.check_output([str(exe)]).split()]
    assert len(errs) == 9 and max(errs) < 5e-6, errs


def test_sinc_limits_host_arithmetic_matches_device_arithmetic():
    """The common filter half size of low_pass / high_pass is computed on the HOST twin of the cutoffs (numpy float32) and
    must equal what the float32 torch formulation -- the device's arithmetic -- gives: a mismatch would size the tap
    table differently from the filters the design kernel writes into it."""
    from audiotools_amd import kernels

    def torch_half_max(c, zeros):
        pos = c > 0
        return int(torch.where(pos, (zeros / torch.where(pos, c, torch.ones_like(c)) / 2).to(torch.int64),
                               torch.zeros_like(c, dtype=torch.int64)).max())

    g = torch.Generator().manual_seed(0)
    for zeros in (51, 8, 24.5, 3):
        for _ in range(300):
            c = torch.rand(17, generator=g) * 0.5
            c[int(torch.randint(0, 17, (1,), generator=g))] = 0.0
            assert kernels._sinc_limits(c, zeros, c) == torch_half_max(c, zeros)
    # values at the rounding edges of zeros / c / 2
    c = torch.tensor([51 / (2 * k) for k in range(52, 450)], dtype=torch.float32)
    assert kernels._sinc_limits(c, 51, c) == torch_half_max(c, 51)
    with pytest.raises(ValueError):
        kernels._sinc_limits(torch.tensor([0.3, 0.51]), 51, torch.tensor([0.3, 0.51]))
    with pytest.raises(ValueError):
        kernels._sinc_limits(torch.tensor([-0.1]), 51, torch.tensor([-0.1]))


def test_paired_inverse_index_plan_replays_irfft():
    """csrc/istft.hip PAIRED: groups j = t, t + L, 3L - t, 4L - t per lane are closed under k -> M - k, the fold needs
    in-lane partners only, and the Stockham passes 4 . 16 . (M / 64) behind the permuted first pass give the transform
    (tools/emulate_istft_paired.py replays the kernel's index arithmetic in numpy)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emulate_istft_paired.py")
    spec = importlib.util.spec_from_file_location("emulate_istft_paired", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for M in (1024, 512, 256):
        fft_err, irfft_err = mod.replay(M, seed=M)
        assert fft_err < 1e-12 and irfft_err < 1e-13, (M, fft_err, irfft_err)


def test_pow2_tile_index_plan_replays_rfft_and_is_conflict_free():
    """csrc/stft_generic.hip stft_tiled_pow2_kernel (n_fft 4096 / 8192): the hand-written LDS slots of the sample stores,
    the three passes and the paired split step give numpy's rfft, every LDS instruction's lane groups hit distinct banks,
    and the lane-level segmented band sums of the mel epilogue equal the dense filterbank product
    (tools/emulate_tiled_pow2.py replays the kernel's index arithmetic in numpy)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emulate_tiled_pow2.py")
    spec = importlib.util.spec_from_file_location("emulate_tiled_pow2", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # the module's own __main__ blocks do not run on import
    for plan in (1, 2):
        mod.run(plan, seed=plan)          # asserts inside
    for sr, n_fft, n_mels, FB, NPC in [(96000, 4096, 80, 2, 4), (96000, 4096, 7, 2, 4), (192000, 8192, 128, 1, 8), (192000, 8192, 3, 1, 8)]:
        mod.mel_reduction(sr, n_fft, n_mels, FB, NPC)


def _load_tool(name):
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_index_plans_replayed_on_the_cpu():
    """The numpy replays the kernels were designed against stay true: the four-step convolution's index plan
    (csrc/longconv.hip) for plans of every shape, the hop-energy pieces of the loudness kernel (csrc/loudness.hip: every hop
    written exactly once, nothing past the end counted), and the fused forward kernel's paired last pass (csrc/stft.hip)."""
    import numpy as np

    lc = _load_tool("emulate_longconv")
    rng = np.random.default_rng(0)
    for T in (16, 120, 2 * 7 * 9, 4096, 9600, 44100):
        x, h = rng.standard_normal(T), rng.standard_normal(T)
        ref = np.fft.irfft(np.fft.rfft(x) * np.fft.rfft(h), T) * 0.37
        y, _ = lc.emulate(T, x, h, 0.37)
        assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-6, T
    _load_tool("emulate_lufs_pieces").main(60)      # asserts inside
    _load_tool("emulate_stft_v2")                   # a script: asserts its error against numpy's rfft on import


def test_build_flags_per_source(monkeypatch):
    """_native.compile_command: no SLP packing for the FFT / filter sources, LLVM's max-ilp scheduler strategy for istft.hip only
    (measured per source, profiles/r03_notes.md), and the A/B switches of the build leave the defaults alone when unset."""
    from audiotools_amd import _native

    for var in ("AT_HIPCC_FLAGS", "AT_STFT_SLP", "AT_NOSLP_ALL", "AT_MAXILP_FILES"):
        monkeypatch.delenv(var, raising=False)
    flags = {s: _native.compile_command("/x/" + s, "/x/o.o") for s in _native.SOURCES}
    ilp = "-amdgpu-sched-strategy=max-ilp"
    assert [s for s, f in flags.items() if ilp in f] == ["istft.hip"]
    for s in ("stft.hip", "istft.hip", "firfft.hip", "longconv.hip", "loudness.hip", "stft_generic.hip"):
        assert "-fno-slp-vectorize" in flags[s]
    for s in ("fir.hip", "irtools.hip", "specedit.hip", "fftconv.hip", "vocoder.hip"):
        assert "-fno-slp-vectorize" not in flags[s]
    assert all("--offload-arch=gfx950" in f and "-O3" in f for f in flags.values())
    # the A/B switches of the build exist in the development build only: the shipped one ignores the environment
    monkeypatch.setenv("AT_MAXILP_FILES", "longconv.hip")
    monkeypatch.setenv("AT_HIPCC_FLAGS", "-DAT_ROW_THREADS=128")
    ship = _native.compile_command("/x/longconv.hip", "/x/o.o")
    assert ilp not in ship and "-DAT_ROW_THREADS=128" not in ship and "-DAT_DEV_KNOBS=1" not in ship
    dev = _native.compile_command("/x/longconv.hip", "/x/o.o", dev=True)
    assert ilp in dev and "-DAT_ROW_THREADS=128" in dev and "-DAT_DEV_KNOBS=1" in dev
    assert ilp not in _native.compile_command("/x/fir.hip", "/x/o.o", dev=True)
    assert _native.compile_command("/x/istft.hip", "/x/o.o", dev=True).count(ilp) == 1


def test_fp16_split_resampler_model_matches_the_oracle():
    """tools/emulate_resample_f16s.py is a lane-level model of resample_f16s_kernel (csrc/resample_f16.hip: tile geometry and
    the 16-byte source alignment shift, clamped prefetch + element-wise re-read of the tiles that reach over an end of the row,
    per-tile power-of-two scale from the exponent field, in-place (hi | lo << 16) split, operand dwords with the K-slot
    permutation, B operands decoded from tables.resample_f16_bank, hh + (hl + lh) in fp32, the store mask).  It asserts its own
    index algebra (every DMA inside the row and aligned, the LDS image equal to the replicate-padded row, every output written
    once); here: same framing and length as julius.resample_frac for several ratios / lengths / row alignments, and closer to
    float64 than the fp32 formulation on loud, quiet, unclipped and 100 dB-dynamic inputs."""
    import numpy as np
    import torch

    mod = _load_tool("emulate_resample_f16s")
    rng = np.random.default_rng(3)
    cases = [(44100, 16000, 5000, 0.1, 0, 2), (44100, 16000, 4410 + 17, 1e-4, 1, 1), (44100, 16000, 7000, 2.0, 3, 3),
             (44100, 16000, 16, 0.5, 2, 1), (44100, 16000, 300, 0.5, 0, 2), (44100, 48000, 2000, 0.3, 1, 2),
             (44100, 24000, 9000, 0.3, 2, 1)]
    for old, new, T, amp, base_word, rows in cases:
        x = (amp * rng.standard_normal((rows, T))).astype(np.float32)
        want = julius_leaf.resample_frac(torch.from_numpy(x)[None].double(), old, new)[0].numpy()
        got = mod.resample(x, old, new, base_word=base_word, n_wg=3)
        assert np.array_equal(got, mod.resample(x, old, new, base_word=base_word, n_wg=2, form="dma"))
        assert got.shape == want.shape
        m = np.abs(want).max()
        f32 = mod.reference(x, old, new, np.float32)
        assert np.abs(got - want).max() / m < 5e-7, (old, new, T)
        assert np.abs(got - want).max() <= 1.5 * np.abs(f32 - want).max() + 1e-9 * m, (old, new, T)
    # a quiet passage next to a loud one inside ONE tile: the error of the quiet half is judged against the quiet level
    T = 6000
    x = np.concatenate([1e-4 * rng.standard_normal(T // 2), rng.standard_normal(T - T // 2)]).astype(np.float32)[None]
    want = julius_leaf.resample_frac(torch.from_numpy(x)[None].double(), 44100, 16000)[0].numpy()
    got = mod.resample(x, 44100, 16000)
    quiet = slice(0, 16000 * (T // 2 - 200) // 44100)
    assert np.abs(got[0, quiet] - want[0, quiet]).max() / np.abs(want[0, quiet]).max() < 2e-6
    # a non-finite sample poisons the outputs whose window holds it -- not the scale of the 16 frames around it
    x = (0.1 * rng.standard_normal((1, 9000))).astype(np.float32)
    x[0, 3000] = np.inf
    got = mod.resample(x, 44100, 16000)
    want = julius_leaf.resample_frac(torch.from_numpy(x)[None].double(), 44100, 16000)[0].numpy()
    clean = np.isfinite(want[0])          # (the reference multiplies the inf with all 581 taps of a frame: a superset)
    assert (~clean).any() and np.isfinite(got[0][clean]).all() and (~np.isfinite(got[0])).any()
    assert np.abs(got[0][clean] - want[0][clean]).max() < 5e-7 * np.abs(want[0][clean]).max()
    centre = 16000 * 3000 // 44100
    assert not np.isfinite(got[0][centre - 5: centre + 5]).any()


def test_fir_adjoint_formula_equals_autograd():
    """filters.fir_adjoint (the backward of the native per-item FIR: the forward kernel on the flipped taps + end corrections
    and folds from prefix sums of the taps) against torch autograd of a float64 replicate-padded correlation: per-item and
    shared taps, high-pass form, L = 1, L = T."""
    import torch
    import torch.nn.functional as F
    from audiotools_amd import filters

    torch.manual_seed(0)

    def fir_cpu(a, table, L):
        B, C, T = a.shape
        H = (L - 1) // 2
        out = torch.empty_like(a)
        for b in range(B):
            k = table[b if table.shape[0] > 1 else 0, :L]
            out[b] = F.conv1d(F.pad(a[b][:, None], (H, H), mode="replicate"), k[None, None])[:, 0]
        return out

    for (B, C, T, L, rows, hp) in [(3, 2, 50, 9, 3, False), (2, 1, 40, 13, 1, True), (2, 2, 31, 31, 2, False), (1, 1, 20, 1, 1, False),
                                   (2, 1, 64, 21, 2, True)]:
        tp = torch.zeros(rows, (L + 7) // 8 * 8, dtype=torch.float64)
        tp[:, :L] = torch.randn(rows, L, dtype=torch.float64)
        x = torch.randn(B, C, T, dtype=torch.float64, requires_grad=True)
        y = fir_cpu(x, tp, L)
        y = (x - y) if hp else y
        g = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, g)
        got = filters.fir_adjoint(g, tp, L, hp, lambda a, t: fir_cpu(a, t, L))
        assert float((got - gx).abs().max()) < 1e-12, (B, C, T, L, rows, hp)


@pytest.mark.parametrize("old_sr,new_sr,T", [(44100, 16000, 1500), (16000, 44100, 400), (2, 1, 50), (1, 2, 50), (3, 2, 64)])
def test_resample_adjoint_bank_is_the_transpose(old_sr, new_sr, T):
    """tables.resample_adjoint_bank: the transposed resampler written as a polyphase bank for at_resample_f32 with the
    rates swapped.  A numpy model of that kernel's sum (groups of 4 phases, dense taps from base[G], replicate padding by
    `width`) on dL/dy with one zero sample on either side, then the fold of the padding, equals autograd's dL/dx of the
    torch formulation in float64."""
    import math
    from audiotools_amd import tables
    g = math.gcd(old_sr, new_sr)
    wg, base, old, new, width, NG, LG, J = tables.resample_adjoint_bank(old_sr // g, new_sr // g)
    assert J * new - 1 >= 1 and (base >= 0).all() and (base <= new + 2 * (J * new - 1)).all()
    rng = np.random.default_rng(1)
    x = torch.tensor(rng.standard_normal(T), dtype=torch.float64, requires_grad=True)
    bank = tables.resample_bank(old_sr, new_sr)[0].double()
    xp = torch.nn.functional.pad(x[None, None], (width, width + old), mode="replicate")
    n = new * T // old
    y = torch.nn.functional.conv1d(xp, bank[:, None], stride=old).transpose(1, 2).reshape(-1)[:n]
    gy = rng.standard_normal(n)
    (y * torch.from_numpy(gy)).sum().backward()
    gin = np.concatenate([[0.0], gy, [0.0]])
    Lp = T + 2 * width + old
    wk = J * new - 1
    frames = (Lp + old - 1) // old
    out = np.zeros(frames * old)
    for f in range(frames):
        for G in range(NG):
            idx = np.clip(f * new + base[G] + np.arange(LG) - wk, 0, len(gin) - 1)
            v = gin[idx] @ wg[:, G, :].astype(np.float64)
            for p in range(4):
                if 4 * G + p < old:
                    out[f * old + 4 * G + p] = v[p]
    out = out[:Lp]
    gx = out[width: width + T].copy()
    gx[0] += out[:width].sum()
    gx[-1] += out[width + T:].sum()
    assert np.abs(gx - x.grad.numpy()).max() < 1e-12 * max(1.0, np.abs(x.grad.numpy()).max())


def test_circular_convolution_adjoints(monkeypatch):
    """fx._NativeCircConv's backward (the forward kernel on index-reversed operands, channel / scale reductions) against
    autograd through the rFFT formulation, with a float64 torch stand-in for the kernel."""
    from audiotools_amd import fx, kernels

    def stand_in(x, ir, scale=None, engine=None):
        T = x.shape[-1]
        y = torch.fft.irfft(torch.fft.rfft(ir, T) * torch.fft.rfft(x, T), T)
        return y if scale is None else y * scale.reshape(ir.shape[0], ir.shape[1], 1)

    monkeypatch.setattr(kernels, "fftconv", stand_in)
    torch.manual_seed(0)
    for B, C, Cir, T in [(3, 2, 1, 50), (3, 2, 2, 51), (2, 1, 1, 64)]:
        x = torch.randn(B, C, T, dtype=torch.float64, requires_grad=True)
        w = torch.randn(B, Cir, T, dtype=torch.float64, requires_grad=True)
        s = (torch.rand(B, Cir, 1, dtype=torch.float64) + 0.5).requires_grad_(True)
        g = torch.randn(B, C, T, dtype=torch.float64)
        got = torch.autograd.grad((fx._NativeCircConv.apply(x, w, s) * g).sum(), (x, w, s))
        ref = torch.autograd.grad((torch.fft.irfft(torch.fft.rfft(w, T) * torch.fft.rfft(x, T), T) * s * g).sum(), (x, w, s))
        for a, b in zip(got, ref):
            assert (a - b).abs().max() < 1e-11


def test_istft_workspace_follows_the_inverse_path():
    """at_istft_workspace_bytes (a host function) sizes the scratch of whichever inverse the shape takes: the one-pass
    kernels need the overlap-add envelope only, the frame-buffer path rows x frames x n_fft floats.  Round 4: run-time sizes
    with an even hop and an even, smooth n_fft / 2 -- speech windows, power-of-two sizes with hops the fused kernels do not
    take -- are one-pass (istft_generic_ola_kernel); odd hops, odd n_fft / 2 and a history deeper than a tile are not."""
    from audiotools_amd import _native
    lib = _native.lib()
    rows, n = 6, 101
    env = lambda n_fft, hop: ((n - 1) * hop + n_fft) * 4
    buf = lambda n_fft: rows * n * n_fft * 4
    for n_fft, hop in [(400, 160), (400, 100), (1200, 300), (1920, 480), (512, 100), (128, 12), (400, 400), (320, 40), (1000, 250)]:
        assert lib.at_istft_workspace_bytes(rows, n, n_fft, hop) == env(n_fft, hop), (n_fft, hop)
    for n_fft, hop in [(100, 33), (882, 441), (1764, 441), (4096, 1000), (1920, 96), (16384, 4096)]:
        assert lib.at_istft_workspace_bytes(rows, n, n_fft, hop) == buf(n_fft), (n_fft, hop)
    # the fused / tiled sizes keep their own (envelope + dump + zero page) layout: at least the envelope, far below the buffer
    for n_fft, hop in [(2048, 512), (512, 128), (128, 8), (4096, 1024), (8192, 2048)]:
        got = lib.at_istft_workspace_bytes(rows, n, n_fft, hop)
        assert env(n_fft, hop) <= got < env(n_fft, hop) + 64 * 1024 + 4 * n_fft * 4, (n_fft, hop, got)


def test_one_pass_inverse_model_vs_torch_istft():
    """tools/emulate_istft_ola.py: tile-level model of istft_generic_ola_kernel's index logic (plan, runs with a warm-up
    tile, history slots in front of the tile, zero spectra outside the row, pair-wise gather in ascending frame order,
    envelope bounds, centre trim) -- every output written exactly once, equal to torch.istft; and its plan agrees with
    the library's (at_istft_workspace_bytes tells which path a shape takes)."""
    from audiotools_amd import _native
    m = _load_tool("emulate_istft_ola")
    lib = _native.lib()
    rng = np.random.default_rng(5)
    for n_fft, hop, T, rows, runs in [(400, 160, 2403, 2, 3), (400, 100, 1500, 1, 2), (1200, 300, 4800, 1, 2), (400, 400, 1200, 1, 1),
                                      (320, 40, 900, 1, 3), (512, 100, 2500, 2, None)]:
        n_frames = 1 + T // hop
        X = (rng.standard_normal((rows, n_frames, n_fft // 2 + 1)) + 1j * rng.standard_normal((rows, n_frames, n_fft // 2 + 1))).astype(np.complex64)
        win = np.hanning(n_fft + 1)[:-1].astype(np.float32) if n_fft % hop == 0 and n_fft // hop >= 2 else np.ones(n_fft, np.float32)
        got = m.istft(X, win, hop, T, force_runs=runs)
        ref = torch.istft(torch.from_numpy(X).transpose(1, 2), n_fft, hop, window=torch.from_numpy(win), center=True, length=T).numpy()
        assert np.abs(got - ref).max() < 2e-6 * np.abs(ref).max(), (n_fft, hop)
    for n_fft, hop in [(400, 160), (1920, 480), (512, 100), (128, 12), (100, 33), (882, 441), (4096, 1000), (1920, 96), (16384, 4096)]:
        one_pass = lib.at_istft_workspace_bytes(3, 50, n_fft, hop) == (49 * hop + n_fft) * 4
        assert one_pass == (m.plan(n_fft, hop) is not None), (n_fft, hop)


def test_long_fir_as_one_circular_convolution(monkeypatch):
    """kernels._fir_long: a per-item FIR of thousands of taps (replicate padding) written as ONE circular convolution at a
    planned length N >= T + L - 1 -- the padded signal, the flipped taps, the rotation by L - 1, the slice, the high-pass
    form, shared taps.  The convolution kernel is replaced by a float64 torch stand-in with the documented semantics of
    at_longconv_room_f32 (IR zero-padded to N and read rotated left by `shift`), so the index algebra is checked on the CPU."""
    from audiotools_amd import kernels

    def stand_in(x, ir, shift=None, scale=None, want_peaks=False):
        B, C, N = x.shape
        L = ir.shape[-1]
        irz = torch.zeros(ir.shape[0], ir.shape[1], N, dtype=torch.float64)
        irz[..., :L] = ir.double()
        if shift is not None:
            idx = (torch.arange(N)[None, None] + shift.reshape(ir.shape[0], ir.shape[1], 1)) % N
            irz = torch.gather(irz, -1, idx)
        return torch.fft.irfft(torch.fft.rfft(irz, N) * torch.fft.rfft(x.double(), N), N).float()

    monkeypatch.setattr(kernels, "room_convolve", stand_in)
    g = torch.Generator().manual_seed(0)
    for B, C, T, L in [(2, 2, 3000, 6145), (3, 1, 5000, 8001), (2, 1, 900, 6201)]:
        x = torch.randn(B, C, T, generator=g)
        taps = torch.randn(B, L, generator=g) / L ** 0.5
        H = (L - 1) // 2
        N = kernels._fir_long_length(T + L - 1)
        assert N >= T + L - 1 and N % 2 == 0 and kernels.longconv_supported(N)
        xp = torch.nn.functional.pad(x.double(), (H, H), mode="replicate")
        ref = torch.stack([torch.nn.functional.conv1d(xp[b][:, None], taps[b].double()[None, None])[:, 0] for b in range(B)])
        Lp = (L + 7) // 8 * 8
        tp = torch.zeros(B, Lp)
        tp[:, :L] = taps
        for highpass in (False, True):
            want = (x.double() - ref) if highpass else ref
            got = kernels._fir_long(x, tp, B, L, highpass, None)
            assert (got.double() - want).abs().max() < 2e-5 * want.abs().max(), (T, L, highpass)
            out = torch.empty_like(x)
            assert kernels._fir_long(x, tp, B, L, highpass, out) is out and torch.equal(out, got)
        shared = kernels._fir_long(x, tp[:1], 1, L, False, None)
        ref0 = torch.nn.functional.conv1d(xp.reshape(-1, 1, xp.shape[-1]), taps[0].double()[None, None]).reshape(B, C, T)
        assert (shared.double() - ref0).abs().max() < 2e-5 * ref0.abs().max()


def test_placed_outputs_pool_logic_on_cpu(monkeypatch):
    """kernels._PlacedOutputs without a GPU: the device queries and events are stand-ins (every "launch" takes as long as the
    candidate's index), the bookkeeping is the real one -- OFF unless opted in; a shape is calibrated at its CALIBRATE_AFTER-th
    call and at most MAX_CALIBRATIONS shapes per process; calibration keeps the fastest KEEP out of as many candidates as
    FREE_FRACTION of the free memory holds; callers get an ALIAS made under the pool's lock, and a buffer is handed out
    again only when nothing references its storage (an alias, a view); the least recently used shape is dropped beyond
    MAX_POOL_BYTES; release() gives everything back."""
    import torch
    from audiotools_amd import kernels

    clock = {"now": 0.0}

    class FakeEvent:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = clock["now"]

        def elapsed_time(self, other):
            return other.t - self.t

    free = {"bytes": 100 * 4096}
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (free["bytes"], 1 << 40))
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    assert kernels._PlacedOutputs.enabled is False and kernels._PlacedOutputs.KEEP == 1, "the pool is an opt-in"
    pool = kernels._PlacedOutputs()
    monkeypatch.setattr(kernels._PlacedOutputs, "MIN_BYTES", 1024)
    made = []

    def alloc():
        t = torch.empty(1024, dtype=torch.float32)
        made.append(t)
        return t

    def launch(t):                       # candidate i "takes" 10 - i ms (the last allocations are the fastest), and writes its result
        i = next(k for k, m in enumerate(made) if m.data_ptr() == t.data_ptr())
        clock["now"] += 10.0 - i
        t.fill_(7.0)

    key = ("cpu", (1024,), 0, "fake")
    assert pool.acquire(key, 4096, alloc, launch) is None and not made, "disabled: never"
    monkeypatch.setattr(kernels._PlacedOutputs, "enabled", True)
    monkeypatch.setattr(kernels._PlacedOutputs, "available", True)
    monkeypatch.setattr(kernels._PlacedOutputs, "KEEP", 3)
    # the first CALIBRATE_AFTER - 1 calls of a shape take plain allocations: a workload of ever-changing shapes never calibrates
    assert kernels._PlacedOutputs.CALIBRATE_AFTER == 3
    assert pool.acquire(key, 4096, alloc, launch) is None and pool.acquire(key, 4096, alloc, launch) is None and not made
    got = pool.acquire(key, 4096, alloc, launch)
    assert got is not None and got[1] is True                      # calibrated: the buffer already holds the result
    rep = pool.report()
    assert rep[0]["op"] == "fake" and len(rep[0]["calibration_ms"]) == kernels._PlacedOutputs.CANDIDATES == len(made)
    assert rep[0]["kept_ms"] == sorted(rep[0]["calibration_ms"])[:3] and rep[0]["bytes_held"] == 3 * 4096 == pool.bytes_held()
    fastest = made[-1]
    assert got[0] is not fastest and got[0].data_ptr() == fastest.data_ptr() and float(got[0][0]) == 7.0, "callers get an alias"
    # the alias made under the lock IS the holder: a second caller (another host thread, before the first has launched) gets the next one
    nxt = pool.acquire(key, 4096, alloc, launch)
    assert nxt[1] is False and nxt[0].data_ptr() == made[-2].data_ptr(), "the fastest buffer is held: the next fastest is handed out"
    view = nxt[0].view(32, 32)                                     # a view holds the storage just the same
    del nxt
    third = pool.acquire(key, 4096, alloc, launch)
    assert third[0].data_ptr() == made[-3].data_ptr()
    assert pool.acquire(key, 4096, alloc, launch) is None, "all three held: plain allocation"
    del got
    assert pool.acquire(key, 4096, alloc, launch)[0].data_ptr() == fastest.data_ptr()    # released -> reused at once
    del view, third
    # as many candidates as FREE_FRACTION of the free memory holds; fewer than two: no pool for the shape
    monkeypatch.setattr(kernels._PlacedOutputs, "CALIBRATE_AFTER", 1)
    free["bytes"] = 11 * 4096
    made.clear()
    assert pool.acquire(("cpu", (1024,), 0, "tight"), 4096, alloc, launch) is not None and len(made) == 5
    free["bytes"] = 3 * 4096
    assert pool.acquire(("cpu", (1024,), 0, "none"), 4096, alloc, launch) is None
    assert pool.acquire(("cpu", (1024,), 0, "none"), 4096, alloc, launch) is None     # remembered
    # least recently used shape dropped beyond MAX_POOL_BYTES
    free["bytes"] = 100 * 4096
    monkeypatch.setattr(kernels._PlacedOutputs, "MAX_POOL_BYTES", 2 * 3 * 4096)
    made.clear()
    pool.acquire(key, 4096, alloc, launch)                                           # touch "fake": "tight" is now the oldest
    made.clear()
    assert pool.acquire(("cpu", (1024,), 0, "third"), 4096, alloc, launch) is not None
    ops = sorted(r["op"] for r in pool.report())
    assert ops == ["fake", "third"], ops
    # at most MAX_CALIBRATIONS calibrations per process ("fake", "tight", "third" so far)
    assert pool.calibrations == 3 and kernels._PlacedOutputs.MAX_CALIBRATIONS == 4
    monkeypatch.setattr(kernels._PlacedOutputs, "MAX_POOL_BYTES", 1 << 30)
    assert pool.acquire(("cpu", (1024,), 0, "fourth"), 4096, alloc, launch) is not None
    made.clear()
    assert pool.acquire(("cpu", (1024,), 0, "fifth"), 4096, alloc, launch) is None and not made
    # release(): everything goes back, the counters start over
    assert pool.bytes_held() > 0
    pool.release()
    assert pool.bytes_held() == 0 and pool.report() == []
    # below the threshold, or disabled: never
    assert pool.acquire(("cpu", (8,), 0, "small"), 512, alloc, launch) is None
    monkeypatch.setattr(kernels._PlacedOutputs, "enabled", False)
    assert pool.acquire(key, 4096, alloc, launch) is None


@pytest.mark.parametrize("M", [16, 32, 64, 128, 256, 512])
def test_run_store_index_algebra(M):
    """Lane-level model of the 512-byte-run stores of the several-frames-per-wave STFT kernels (csrc/stft.hip, RUNSTORE):
    the split step's lanes (frame slot fs = lane / L, t = lane % L) drop bins k = t + L q and M - k (q < 8; lane t = 0 also
    M / 2) into the slab at fs * 17 L + bin; the wave then reads the slab back by the LINEAR output index idx = 64 i + lane
    (frame idx / (M + 1), bin idx % (M + 1)) in 17 instructions.  Every (frame, bin) must be written exactly once, fit the
    1088-slot slab, be read back at its own output offset, and a ragged last group must store exactly the frames that exist."""
    L = M // 16
    FW = 64 // L
    SLOTS = 17 * L
    slab = {}
    for lane in range(64):
        fs, t = lane // L, lane % L
        for q in range(8):
            k = t + L * q
            for b in (k, M - k):
                slot = fs * SLOTS + b
                assert slot not in slab and slot < 1088
                slab[slot] = (fs, b)
        if t == 0:
            slot = fs * SLOTS + M // 2
            assert slot not in slab
            slab[slot] = (fs, M // 2)
    TOT = FW * (M + 1)
    NST = (TOT + 63) // 64
    assert len(slab) == TOT and NST == 17
    for live in range(1, FW + 1):                       # frames of the group that exist
        tot = live * (M + 1)
        stored = {}
        for i in range(NST):
            for lane in range(64):
                if live == FW:
                    idx = min(i * 64 + lane, TOT - 1)   # full group: unconditional stores, the last instruction clamps
                elif i * 64 + lane < tot:
                    idx = i * 64 + lane
                else:
                    continue
                f = idx // (M + 1)
                k = idx - f * (M + 1)
                assert slab[f * SLOTS + k] == (f, k)
                stored[idx] = (f, k)                    # (duplicates of the clamped lanes carry the same value)
        assert sorted(stored) == list(range(tot))


def test_ola_envelope_equals_the_transposed_convolution():
    """kernels._ola_envelope (one index_add_ in float64) is the overlap-add envelope torch.istft divides by; the adjoints used
    conv_transpose1d for it until round 6 -- a MIOpen call that aborts the interpreter when it is a process's first one and comes
    from autograd's worker thread (profiles/r06_notes.md 5.2)."""
    import torch
    from audiotools_amd import kernels

    for n_fft, hop, nf in [(2048, 512, 40), (512, 128, 60), (1024, 512, 21), (256, 32, 90), (64, 4, 120), (400, 160, 33), (2048, 700, 9), (64, 64, 1)]:
        w = torch.hann_window(n_fft) + 0.01
        ref = torch.nn.functional.conv_transpose1d(torch.ones(1, 1, nf), (w ** 2)[None, None], stride=hop)[0, 0]
        got = kernels._ola_envelope(w, hop, nf)
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), (n_fft, hop, nf)
