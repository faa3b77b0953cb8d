// In-place edits of stft_data between stft() and istft() (gfx950).
//
// Replaces the polar round trips of reference audiotools/core/dsp.py:217-370
//     mag, phase = abs(X), angle(X); masked_fill / + shift; X = mag * exp(1j * phase)
// (8-10 whole-tensor passes over a complex spectrogram each) by single kernels on the physical
// (rows, frames, bins) complex64 layout:
//   at_spec_mask_f32          dsp.py:217-306  mask_frequencies / mask_timesteps
//   at_spec_phase_shift_f32   dsp.py:336-352  shift_phase with one shift per item
//   at_spec_maxpow_f32 +
//   at_spec_mask_lowmag_f32   dsp.py:308-334  mask_low_magnitudes (log_magnitude incl. the global
//                                             top_db floor of audio_signal.py:1457-1487)
// Every kernel works out of place (src -> X: the reference returns a NEW stft_data, and reading src
// while writing X costs one pass instead of clone + edit) or in place (src NULL or == X: only the
// masked region is written).  Unmasked elements are copied / left untouched (the reference rewrites
// them as |X| e^{i angle X}, which is X up to rounding).
#include "at_common.h"

namespace {

// one wave per (row, frame); lanes stride over the bins
__global__ __launch_bounds__(256) void spec_mask_kernel(const float2* __restrict__ S, float2* __restrict__ X, int64_t rows,
                                                        int C, int N, int F, int axis, const double* __restrict__ lo,
                                                        const double* __restrict__ hi, const float* __restrict__ grid,
                                                        float2 fill) {
  const int lane = threadIdx.x & 63;
  const int64_t total = rows * N;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += (int64_t)gridDim.x * 4) {
    const int64_t row = w / N;
    const int n = (int)(w - row * N);
    const int64_t b = row / C;
    const double l = lo[b], h = hi[b];
    float2* __restrict__ Xf = X + w * F;
    if (S) {  // out of place: every element is written (copied or filled)
      const float2* __restrict__ Sf = S + w * F;
      const bool whole = axis == 1 && l <= (double)grid[n] && (double)grid[n] < h;
      for (int f = lane; f < F; f += 64) {
        const double g = (double)grid[axis == 1 ? n : f];
        const bool m = axis == 1 ? whole : (l <= g && g < h);
        Xf[f] = m ? fill : Sf[f];
      }
    } else if (axis == 1) {  // in place, time: the whole frame or nothing
      const double g = (double)grid[n];
      if (!(l <= g && g < h)) continue;
      for (int f = lane; f < F; f += 64) Xf[f] = fill;
    } else {
      for (int f = lane; f < F; f += 64) {
        const double g = (double)grid[f];
        if (l <= g && g < h) Xf[f] = fill;
      }
    }
  }
}

__global__ __launch_bounds__(256) void spec_phase_shift_kernel(const float2* __restrict__ S, float2* __restrict__ X,
                                                               int64_t rows, int C, int64_t per_row,
                                                               const float* __restrict__ shift) {
  const int64_t row = blockIdx.y;
  const float sh = shift[row / C];
  float sn, cs;
  sincosf(sh, &sn, &cs);
  float2* __restrict__ Xr = X + row * per_row;
  const float2* __restrict__ Sr = (S ? S : X) + row * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_row; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 v = Sr[i];
    Xr[i] = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
  }
}

// max over all elements of |X|^2 as the reference forms it: fl(fl(|X|)^2).  Non-negative floats
// order like their bit patterns, so the cross-block reduction is an integer atomicMax.
__global__ __launch_bounds__(256) void spec_maxpow_kernel(const float2* __restrict__ X, int64_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 v = X[i];
    const float mag = hypotf(v.x, v.y);
    m = fmaxf(m, mag * mag);
  }
  m = at::wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void spec_mask_lowmag_kernel(const float2* __restrict__ S, float2* __restrict__ X, int64_t rows,
                                                               int C, int64_t per_row,
                                                               const double* __restrict__ cutoff_db,
                                                               const unsigned* __restrict__ maxpow, float top_db,
                                                               int use_top_db, float val) {
  const int64_t row = blockIdx.y;
  const double cut = cutoff_db[row / C];
  const float amin2 = 1e-10f;  // amin ** 2 = 1e-10 (python float) applied to a float32 tensor
  // log_spec.max() - top_db
  const float floor_db = 10.0f * log10f(fmaxf(__uint_as_float(*maxpow), amin2)) - top_db;
  float2* __restrict__ Xr = X + row * per_row;
  const float2* __restrict__ Sr = (S ? S : X) + row * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_row; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 v = Sr[i];
    const float mag = hypotf(v.x, v.y);
    float ls = 10.0f * log10f(fmaxf(mag * mag, amin2));
    if (use_top_db) ls = fmaxf(ls, floor_db);
    if ((double)ls < cut) {
      // magnitude := val, phase kept: val * e^{i angle(X)}  (angle(0) = 0)
      const float inv = mag > 0.f ? 1.0f / mag : 0.f;
      Xr[i] = mag > 0.f ? make_float2(val * v.x * inv, val * v.y * inv) : make_float2(val, 0.f);
    } else if (S) {
      Xr[i] = v;
    }
  }
}


// ---- per-ELEMENT polar edits: corrupt_phase / CorruptPhase, TimeNoise / FrequencyNoise --------
// The per-element operands arrive in the LOGICAL layout (rows, F, N) (they are made by
// torch.randn_like(phase) or come from the host as a (B, C, F, N) array), the spectrum is
// physically (rows, N, F): a 32 x 32 tile goes through LDS so that both sides are read / written
// with unit stride.
//   mode 0: Y = X * e^{i shift}                          dsp.py:354-370, transforms.py:1250-1278
//   mode 1: Y = (X == 0) ? |a| * e^{i b} : X             transforms.py:1456-1536 (the holes a mask left
//           are refilled with magnitude |N(0,1)| -- `signal.magnitude = mag; signal.phase = phase` ends with
//           abs(mag) e^{i phase} -- and phase ~ N(0,1); untouched bins keep their value)
__global__ __launch_bounds__(256) void spec_polar_elem_kernel(const float2* __restrict__ S, float2* __restrict__ Y, int N,
                                                              int F, const float* __restrict__ a, const float* __restrict__ b,
                                                              int mode) {
  __shared__ float2 tile[32][33];
  const int64_t row = blockIdx.z;
  const int f0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* __restrict__ ar = a ? a + row * (int64_t)F * N : nullptr;
  const float* __restrict__ br = b + row * (int64_t)F * N;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + ty + 8 * r, n = n0 + tx;
    float2 v = make_float2(0.f, 0.f);
    if (f < F && n < N) {
      float sn, cs;
      sincosf(br[(int64_t)f * N + n], &sn, &cs);
      const float m = mode == 1 ? fabsf(ar[(int64_t)f * N + n]) : 1.0f;   // the phase setter re-derives |.| (audio_signal.py:1452-1453)
      v = make_float2(m * cs, m * sn);
    }
    tile[ty + 8 * r][tx] = v;      // tile[f][n]
  }
  __syncthreads();
  const float2* __restrict__ Sr = S + row * (int64_t)N * F;
  float2* __restrict__ Yr = Y + row * (int64_t)N * F;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + ty + 8 * r, f = f0 + tx;
    if (f < F && n < N) {
      const float2 x = Sr[(int64_t)n * F + f];
      const float2 e = tile[tx][ty + 8 * r];
      float2 y;
      if (mode == 0) y = make_float2(x.x * e.x - x.y * e.y, x.x * e.y + x.y * e.x);
      else y = (x.x == 0.f && x.y == 0.f) ? e : x;
      Yr[(int64_t)n * F + f] = y;
    }
  }
}

// ---- spectral gate (ml/layers/spectral_gate.py:58-127) -----------------------------------------------
// gate[f, n] = 20 log10(max(|X|, 1e-4)) < thr_db[f];  S = conv2d(gate, outer(tf, tt), zero padding);
// Y = X (1 - amount S).  The reference runs ~8 whole-tensor passes (abs, clamp, log10, compare, float, conv2d,
// scale, complex multiply); here one workgroup owns 16 frames of one row: the gate bits of the 16 + 2 halo_t
// frames go to LDS as bytes, every thread then owns a column (bin) -- the smoothing is separable, so the tent
// over frequency is applied per frame from 7 byte reads into registers and the tent over time runs on those
// registers -- and the product is written.  X is read once from HBM (1.6x with the halo frames, the second
// touch of the 16 centre frames comes from L2), Y written once.
constexpr int GATE_TN = 16;         // frames per workgroup
constexpr int GATE_MAXH = 8;        // largest half width of either tent

struct GateKArgs {
  const float2* X;       // (rows, N, F)
  float2* Y;
  const float* thr_db;   // (thr_rows, F)
  const float* amount;   // (B)
  const float* tf;       // (kf) tent over frequency, already divided by the sum of the 2-D kernel
  const float* tt;       // (kt) tent over time
  int64_t rows;
  int N, F, C, kf, kt, thr_per_item;   // thr_per_item: thr_db has one row per signal row, else one per channel
};

__global__ __launch_bounds__(256) void spec_gate_kernel(const GateKArgs A) {
  // dynamic LDS only (a static array next to a raised dynamic limit fails the launch): the two tents, then the bits
  extern __shared__ __attribute__((aligned(16))) unsigned char gate_lds[];
  float* s_tf = reinterpret_cast<float*>(gate_lds);
  float* s_tt = s_tf + 2 * GATE_MAXH + 1;
  unsigned char* gbits = gate_lds + 2 * (2 * GATE_MAXH + 1) * sizeof(float) + 8;   // [TN + 2 ht][F + 2 hf]
  const int hf = A.kf / 2, ht = A.kt / 2;
  const int FP = A.F + 2 * hf;
  const int tiles = (A.N + GATE_TN - 1) / GATE_TN;
  const int64_t row = blockIdx.x / tiles;
  const int n0 = (int)(blockIdx.x - row * tiles) * GATE_TN;
  const int nrows = GATE_TN + 2 * ht;
  if (threadIdx.x < A.kf) s_tf[threadIdx.x] = A.tf[threadIdx.x];
  if (threadIdx.x < A.kt) s_tt[threadIdx.x] = A.tt[threadIdx.x];
  const float2* __restrict__ Xr = A.X + row * (int64_t)A.N * A.F;
  const float* __restrict__ thr = A.thr_db + (A.thr_per_item ? row : row % A.C) * (int64_t)A.F;
  // gate bits of frames n0 - ht .. n0 + TN + ht - 1 (zeros outside the spectrogram and in the frequency halo)
  for (int i = threadIdx.x; i < nrows * FP; i += 256) {
    const int r = i / FP, fp = i - r * FP;
    const int n = n0 - ht + r, f = fp - hf;
    unsigned char g = 0;
    if (n >= 0 && n < A.N && f >= 0 && f < A.F) {
      const float2 v = Xr[(int64_t)n * A.F + f];
      const float db = 20.0f * log10f(fmaxf(hypotf(v.x, v.y), 1e-4f));
      g = db < thr[f] ? 1 : 0;
    }
    gbits[i] = g;
  }
  __syncthreads();
  const float amt = A.amount[row / A.C];
  float2* __restrict__ Yr = A.Y + row * (int64_t)A.N * A.F;
  for (int f = threadIdx.x; f < A.F; f += 256) {
    float g1[GATE_TN + 2 * GATE_MAXH];
#pragma unroll
    for (int r = 0; r < GATE_TN + 2 * GATE_MAXH; ++r) {
      float acc = 0.f;
      if (r < nrows) {
        const unsigned char* __restrict__ gp = gbits + r * FP + f;      // taps f - hf .. f + hf sit at f .. f + 2 hf
        for (int d = 0; d < A.kf; ++d) acc = fmaf(s_tf[d], (float)gp[d], acc);
      }
      g1[r] = acc;
    }
#pragma unroll
    for (int j = 0; j < GATE_TN; ++j) {
      const int n = n0 + j;
      if (n < A.N) {
        float sm = 0.f;
#pragma unroll
        for (int d = 0; d < 2 * GATE_MAXH + 1; ++d)
          if (d < A.kt) sm = fmaf(s_tt[d], g1[j + d], sm);
        const float keep = 1.0f - amt * sm;
        const float2 v = Xr[(int64_t)n * A.F + f];
        Yr[(int64_t)n * A.F + f] = make_float2(v.x * keep, v.y * keep);
      }
    }
  }
}

}  // namespace

extern "C" {

// Spectral gate: Y = X (1 - amount[b] * conv2d(gate, outer(tf, tt))) with gate = 20 log10(max(|X|, 1e-4)) < thr_db
// (ml/layers/spectral_gate.py:96-121).  X, Y (B, C, N, F) complex64 bin-contiguous; thr_db (B * C, F) when
// thr_per_item else (C, F) (a one-item noise clip broadcast over the batch); tf (kf), tt (kt): the two tents, their
// outer product being the NORMALISED smoothing filter (kf, kt odd, <= 17); amount (B).
int at_spec_gate_f32(const float* X, float* Y, int64_t B, int64_t C, int64_t N, int64_t F, const float* thr_db,
                     int thr_per_item, const float* amount, const float* tf, int kf, const float* tt, int kt, void* stream) {
  if (B == 0) return AT_OK;
  if (!X || !Y || !thr_db || !amount || !tf || !tt || B < 0 || C <= 0 || N <= 0 || F <= 0) return AT_ERR_INVALID;
  if (kf < 1 || kt < 1 || !(kf & 1) || !(kt & 1) || kf > 2 * GATE_MAXH + 1 || kt > 2 * GATE_MAXH + 1) return AT_ERR_UNSUPPORTED;
  if (N >= (1LL << 31) || F >= (1LL << 30)) return AT_ERR_UNSUPPORTED;
  GateKArgs A;
  A.X = reinterpret_cast<const float2*>(X); A.Y = reinterpret_cast<float2*>(Y); A.thr_db = thr_db; A.amount = amount;
  A.tf = tf; A.tt = tt; A.rows = B * C; A.N = (int)N; A.F = (int)F; A.C = (int)C; A.kf = kf; A.kt = kt;
  A.thr_per_item = thr_per_item;
  const size_t lds = (size_t)(GATE_TN + 2 * (kt / 2)) * (F + 2 * (kf / 2)) + 2 * (2 * GATE_MAXH + 1) * sizeof(float) + 8;
  if (lds > 150 * 1024) return AT_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {
    int e = at::allow_big_lds(reinterpret_cast<const void*>(spec_gate_kernel));
    if (e != AT_OK) return e;
  }
  const int64_t tiles = (N + GATE_TN - 1) / GATE_TN;
  if (B * C * tiles > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(spec_gate_kernel, dim3((unsigned)(B * C * tiles)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// X (B, C, N, F) complex64 interleaved, bin-contiguous.  axis 0: bins f with lo[b] <= grid[f] < hi[b]
// (grid = linspace(0, sr/2, F) as float32); axis 1: frames n with lo[b] <= grid[n] < hi[b]
// (grid = linspace(0, duration, N)).  Masked elements become fill = (re, im) = val * e^{i val}.
int at_spec_mask_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, int axis, const double* lo,
                     const double* hi, const float* grid, float fill_re, float fill_im, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!X || !lo || !hi || !grid || B < 0 || C <= 0 || N <= 0 || F <= 0 || (axis != 0 && axis != 1) || N >= (1LL << 31) ||
      F >= (1LL << 31))
    return AT_ERR_INVALID;
  if (B == 0) return AT_OK;
  const int64_t waves = B * C * N;
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(spec_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float2*>(src == X ? nullptr : src), reinterpret_cast<float2*>(X), B * C, (int)C, (int)N,
                     (int)F, axis, lo, hi, grid, make_float2(fill_re, fill_im));
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// X *= e^{i shift[b]}  (shift_phase with one value per item)
int at_spec_phase_shift_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, const float* shift,
                            void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!X || !shift || B < 0 || C <= 0 || N <= 0 || F <= 0) return AT_ERR_INVALID;
  if (B == 0) return AT_OK;
  const int64_t rows = B * C, per_row = N * F;
  if (rows > 65535) return AT_ERR_UNSUPPORTED;
  int64_t bx = (per_row + 256 * 8 - 1) / (256 * 8);
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(spec_phase_shift_kernel, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(src == X ? nullptr : src),
                     reinterpret_cast<float2*>(X), rows, (int)C, per_row, shift);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// *out (device, float) = max |X|^2 over n complex elements
int at_spec_maxpow_f32(const float* X, int64_t n, float* out, void* stream) {
  if (!X || !out || n < 0) return AT_ERR_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(out, 0, 4, st);
  if (e != hipSuccess) return AT_ERR_HIP(e);
  if (n == 0) return AT_OK;
  int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(spec_maxpow_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float2*>(X), n,
                     reinterpret_cast<unsigned*>(out));
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// mask_low_magnitudes: elements whose log-magnitude (10 log10 max(|X|^2, 1e-10), floored at the
// global maximum - top_db when use_top_db) is below cutoff_db[b] get magnitude `val`, phase kept.
// maxpow: device float from at_spec_maxpow_f32 of the same tensor.
int at_spec_mask_lowmag_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, const double* cutoff_db,
                            const float* maxpow, float top_db, int use_top_db, float val, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!X || !cutoff_db || !maxpow || B < 0 || C <= 0 || N <= 0 || F <= 0) return AT_ERR_INVALID;
  if (B == 0) return AT_OK;
  const int64_t rows = B * C, per_row = N * F;
  if (rows > 65535) return AT_ERR_UNSUPPORTED;
  int64_t bx = (per_row + 256 * 8 - 1) / (256 * 8);
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(spec_mask_lowmag_kernel, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(src == X ? nullptr : src),
                     reinterpret_cast<float2*>(X), rows, (int)C, per_row, cutoff_db, reinterpret_cast<const unsigned*>(maxpow), top_db,
                     use_top_db, val);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// Per-element polar edits of stft_data.  src / X: (B, C, N, F) complex64 (src may equal X or be NULL
// for in place); a, b: (B, C, F, N) float32 in the LOGICAL layout of the reference's tensors.
//   mode 0: X = src * e^{i b}  (a unused, may be NULL)      shift_phase / corrupt_phase with a full tensor
//   mode 1: X = (src == 0) ? a e^{i b} : src                TimeNoise / FrequencyNoise refill
int at_spec_polar_elem_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, const float* a,
                           const float* b, int mode, void* stream) {
  if (B == 0) return AT_OK;
  if (!X || !b || B < 0 || C <= 0 || N <= 0 || F <= 0 || (mode != 0 && mode != 1) || (mode == 1 && !a)) return AT_ERR_INVALID;
  const int64_t rows = B * C;
  if (N >= (1LL << 31) || F >= (1LL << 31) || (F + 31) / 32 > 65535) return AT_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float2* S = reinterpret_cast<const float2*>(src ? src : X);
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid((unsigned)((N + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)nr);
    hipLaunchKernelGGL(spec_polar_elem_kernel, grid, dim3(256), 0, st, S + r0 * N * F, reinterpret_cast<float2*>(X) + r0 * N * F,
                       (int)N, (int)F, a ? a + r0 * F * N : nullptr, b + r0 * F * N, mode);
    AT_LAUNCH_CHECK();
  }
  return AT_OK;
}


}  // extern "C"
