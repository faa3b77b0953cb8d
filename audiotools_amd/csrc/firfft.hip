// Overlap-save FFT form of the per-item FIR (gfx950).
//
// Same contract as fir_per_item_kernel in fir.hip (reference audiotools/core/dsp.py:177-179,
// 209-211 julius Low/HighPassFilter per item; core/effects.py:399-403, 429-432 julius SplitBands
// + weighted band sum collapsed to one composite FIR per item):
//     y[n] = sum_j h[j] x[clamp(n - half + j)],   out = y  or  x - y (highpass)
// but the cost per output no longer grows with the tap count: direct evaluation spends
// 2*taps flop per sample (1047 taps for the 6-band equalizer at 48 kHz: 36 % of the FP32 vector
// peak and still 14 ms at cfg4), the block FFT spends ~200.
//
// One wave owns a run of consecutive blocks of one (item, channel) row:
//   setup   G = conj(FFT_N(h)) / (2N), kept in registers as the pair tables S, D  (N = 2048)
//   block   u = x[s .. s+N) (replicate clamp), z[n] = u[2n] + i u[2n+1], Z = FFT_M(z), M = N/2
//           with A = Z[k], B = conj Z[M-k], w = e^{-2 pi i k/N}:
//               E = A + B, O = -i w (A - B)           (U[k] = (E+O)/2, U[M-k] = conj(E-O)/2)
//               P = E S + O D,  Q = E D + O S         (S,D = (G[k] +- conj G[M-k]) / 2N)
//               Z'[k] = P + i conj(w) Q,  Z'[M-k] = conj(P - i conj(w) Q)
//           z' = conj(FFT_M(conj Z')):  y[s + half + 2n] = Re z'[n], y[.. + 1] = -Im ... the
//           first V = N - taps + 1 samples of the circular correlation are exact.
// The wave FFT (fft_wave.h) is the one of the STFT kernel; waves never synchronise with each
// other.  Taps longer than FIRFFT_MAX_PART are cut into partitions of FIRFFT_PART taps whose
// outputs accumulate (uniformly partitioned overlap-save), one launch per partition.
#include "at_common.h"
#include "fft_wave.h"

namespace {

constexpr int FF_M = 1024;            // complex FFT length
constexpr int FF_N = 2 * FF_M;        // block length in samples
constexpr int FF_NW = 4;              // waves per workgroup
#ifndef FF_WPS
#define FF_WPS 2                      // resident waves per SIMD the register budget allows
#endif

struct FirFftArgs {
  const float* x;      // (rows, T)
  const float* taps;   // (taps_rows, Lp)
  const float2* tw;    // (N): (cos, -sin)(2 pi k / N)
  float* out;          // (rows, T)
  int64_t T;
  int64_t rows;
  int C;
  int taps_rows;
  int Lp;              // row pitch of taps
  int j0, nj;          // partition: taps [j0, j0 + nj)
  int half;            // centre tap index (whole filter)
  int highpass;        // out = x - y; the delta sits in the partition that contains `half`
  int accumulate;      // out += instead of out =
  int V;               // valid outputs per block (even)
  int blocks_per_row;
  int nseg;            // runs per row
  int blocks_per_seg;
};

struct __attribute__((packed, aligned(4))) f2u { float x, y; };  // dword-aligned pair

__global__ __launch_bounds__(FF_NW * 64, FF_WPS) void fir_fft_kernel(const FirFftArgs A) {
  constexpr int M = FF_M, N = FF_N, L = 64;
  __shared__ float2 lds[FF_NW * WAVE_LDS_SLOTS];
  __shared__ float2 s_twp[M / 2 + 1];
  __shared__ __attribute__((aligned(16))) float s_tw2[16 * 36];
  for (int i = threadIdx.x; i <= M / 2; i += FF_NW * 64) s_twp[i] = A.tw[i];
  for (int i = threadIdx.x; i < 16 * 16; i += FF_NW * 64) {
    const int jj = i / 16, r = i % 16;
    reinterpret_cast<float2*>(s_tw2 + jj * 36)[r] = A.tw[r * jj * (N / 256)];
  }
  __syncthreads();

  const int t = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2* fbuf = lds + wave * WAVE_LDS_SLOTS;

  const int64_t unit = (int64_t)blockIdx.x * FF_NW + wave;   // (row, segment)
  if (unit >= A.rows * A.nseg) return;                        // wave-uniform; no barriers below
  const int64_t row = unit / A.nseg;
  const int seg = (int)(unit - row * A.nseg);
  const int64_t item = row / A.C;
  const float* __restrict__ xr = A.x + row * A.T;
  float* __restrict__ orow = A.out + row * A.T;
  const float* __restrict__ h = A.taps + (A.taps_rows == 1 ? 0 : item) * (int64_t)A.Lp + A.j0;

  float2 tw3b[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) tw3b[b] = A.tw[((t + b * L) % 256) * (N / 1024)];

  // forward M-point FFT of a[] (a[q] = point t + L q); the result is left in fbuf (natural order)
  auto fft = [&](float2 (&a)[16]) __attribute__((always_inline)) {
    pass_compute_store<16, 1, L>(a, fbuf, t, nullptr);
    wave_sync();
    load_points<L>(a, fbuf, t);
    wave_sync();
    {
      float2 tw2[16];
      const float2* rowp = reinterpret_cast<const float2*>(s_tw2 + (t & 15) * 36);
#pragma unroll
      for (int r = 1; r < 16; ++r) tw2[r] = rowp[r];
      pass_compute_store<16, 16, L>(a, fbuf, t, tw2);
    }
    wave_sync();
    load_points<L>(a, fbuf, t);
    wave_sync();
    {
      float2 tw3[16];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        tw3[b * 4 + 1] = tw3b[b];
        tw3[b * 4 + 2] = cmul(tw3b[b], tw3b[b]);
        tw3[b * 4 + 3] = cmul(tw3[b * 4 + 2], tw3b[b]);
      }
      pass_compute_store<4, 256, L>(a, fbuf, t, tw3);
    }
    wave_sync();
  };

  // ---- setup: pair tables of the filter partition
  float2 S[8], D[8];
  float2 Gh;  // conj(Hf[M/2]) * 2/N, used by lane 0
  {
    float2 a[16];
    // position of the highpass delta inside this partition (-1: it belongs to another one)
    const int dpos = (A.half >= A.j0 && A.half < A.j0 + A.nj) ? A.half - A.j0 : -1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = 2 * (t + L * q);
      float h0 = m < A.nj ? h[m] : 0.f;
      float h1 = m + 1 < A.nj ? h[m + 1] : 0.f;
      if (A.highpass) {
        h0 = (m == dpos ? 1.f : 0.f) - h0;
        h1 = (m + 1 == dpos ? 1.f : 0.f) - h1;
      }
      a[q] = make_float2(h0, h1);
    }
    fft(a);
    const float sc = 1.0f / (float)(2 * N);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = t + L * q;
      const float2 zk = fbuf[phys<L>(k)];
      const float2 zm = fbuf[phys<L>((M - k) & (M - 1))];
      float2 hk, hm;  // Hf[k], Hf[M-k]
      if (q == 0 && t == 0) {
        hk = make_float2(zk.x + zk.y, 0.f);
        hm = make_float2(zk.x - zk.y, 0.f);
      } else {
        const float2 w = s_twp[k];
        const float sr = zk.x + zm.x, si = zk.y - zm.y;
        const float dr = zk.x - zm.x, di = zk.y + zm.y;
        const float c = w.x, s = -w.y;
        const float pp = fmaf(s, dr, -c * di);
        const float qq = fmaf(s, di, c * dr);
        hk = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));
        hm = make_float2(0.5f * (sr + pp), 0.5f * (-si - qq));
      }
      // G[k] = conj(Hf[k]);  Gm = conj(G[M-k]) = Hf[M-k]
      const float2 gk = make_float2(hk.x, -hk.y), gm = hm;
      S[q] = make_float2((gk.x + gm.x) * sc, (gk.y + gm.y) * sc);
      D[q] = make_float2((gk.x - gm.x) * sc, (gk.y - gm.y) * sc);
    }
    {
      const float2 zh = fbuf[phys<L>(M / 2)];  // Hf[M/2] = conj(zh) -> G[M/2] = zh
      Gh = make_float2(zh.x * (2.0f / (float)N), zh.y * (2.0f / (float)N));
    }
    wave_sync();
  }

  const int V = A.V;
  const int Ti = (int)A.T;
  const int b_lo = seg * A.blocks_per_seg;
  const int b_hi = min(b_lo + A.blocks_per_seg, A.blocks_per_row);
  // Input of a block: interior blocks take 16 dword-aligned 8-B loads per lane, issued one block
  // AHEAD (right after pass 1 of the current block consumed the registers); row-edge blocks are
  // staged with replicate clamping through the wave's slab at the top of their own iteration.
  auto interior = [&](int blk) { const int s0 = blk * V - A.half + A.j0; return s0 >= 0 && s0 + N <= Ti; };
  float2 nx[16];
  auto issue_loads = [&](int blk) __attribute__((always_inline)) {
    const float* __restrict__ p = xr + (blk * V - A.half + A.j0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const f2u v = *reinterpret_cast<const f2u*>(p + 2 * (t + L * q));
      nx[q] = make_float2(v.x, v.y);
    }
  };
  bool have_nx = false;
  if (b_lo < b_hi && interior(b_lo)) { issue_loads(b_lo); have_nx = true; }
  const int src_lane = (64 - t) & 63;

  for (int blk = b_lo; blk < b_hi; ++blk) {
    const int o0 = blk * V;                 // first output of the block (T < 2^31 checked by the host)
    const int s0 = o0 - A.half + A.j0;      // first input sample
    float2 a[16];
    if (have_nx) {                          // wave-uniform
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = nx[q];
    } else {
      // row edges: replicate-clamped samples staged through the wave's slab (linear floats)
      float* stage = reinterpret_cast<float*>(fbuf);
#pragma unroll 1
      for (int m = t; m < N; m += 64) {
        int g = s0 + m;
        g = g < 0 ? 0 : (g >= Ti ? Ti - 1 : g);
        stage[m] = xr[g];
      }
      wave_sync();
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = reinterpret_cast<const float2*>(stage)[t + L * q];
      wave_sync();
    }
    have_nx = blk + 1 < b_hi && interior(blk + 1);
    if (have_nx) issue_loads(blk + 1);
    fft(a);
    // ---- spectrum product on the packed form.  Lane t produces conj(Z') at k = t + 64 q (q < 8),
    // which IS point t + 64 q of the inverse transform's first pass, and at M - k, which is point
    // (64 - t) + 64 (15 - q): register 15 - q of lane (64 - t) mod 64 -- handed over with
    // ds_bpermute instead of a round trip through the slab (lane 0 pairs with itself).
    float2 cmv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = t + L * q;
      const float2 zk = fbuf[phys<L>(k)];
      const float2 zm = fbuf[phys<L>((M - k) & (M - 1))];
      const float2 w = s_twp[k];
      const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
      const float dr = zk.x - zm.x, di = zk.y + zm.y;
      const float2 O = make_float2(fmaf(w.x, di, w.y * dr), -fmaf(w.x, dr, -w.y * di));
      const float2 Pp = cadd(cmul(E, S[q]), cmul(O, D[q]));
      const float2 Qq = cadd(cmul(E, D[q]), cmul(O, S[q]));
      // R = i conj(w) Q
      const float2 R = make_float2(-fmaf(w.x, Qq.y, -w.y * Qq.x), fmaf(w.x, Qq.x, w.y * Qq.y));
      a[q] = make_float2(Pp.x + R.x, -(Pp.y + R.y));   // conj(Z'[k]) = conj(P + R)
      cmv[q] = make_float2(Pp.x - R.x, Pp.y - R.y);    // conj(Z'[M-k]) = P - R
    }
    // k == M/2 (point 512 = register 8 of lane 0): conj(Z') = (2/N) conj(Z[M/2]) G[M/2]
    const float2 zmid = fbuf[phys<L>(M / 2)];
    const float2 mid = cmul(make_float2(zmid.x, -zmid.y), Gh);
    wave_sync();  // all reads of Z done before pass 1 of the inverse overwrites the slab
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 sv = make_float2(__shfl(cmv[7 - j].x, src_lane, 64), __shfl(cmv[7 - j].y, src_lane, 64));
      const float2 own = j == 0 ? mid : cmv[j == 0 ? 0 : 8 - j];
      a[8 + j] = t == 0 ? own : sv;
    }
    fft(a);
    // ---- z'[n] = conj(F[n]): outputs o0 + 2n, o0 + 2n + 1 for 2n < lim
    const int lim = min(Ti - o0, V);
    // all sixteen slab reads first: inside the (wave-uniform) range tests below every q was its own basic block,
    // i.e. sixteen LDS round trips one after the other
    float2 yv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) yv[q] = fbuf[phys<L>(t + L * q)];
    if (!A.accumulate) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = t + L * q;
        const float2 y = yv[q];
        float* __restrict__ po = orow + o0 + 2 * n;
        if (2 * L * (q + 1) <= lim) {            // the whole wave is inside: wave-uniform, no exec masking
          f2u o; o.x = y.x; o.y = -y.y;
          *reinterpret_cast<f2u*>(po) = o;
        } else if (2 * L * q < lim) {
          if (2 * n < lim) po[0] = y.x;
          if (2 * n + 1 < lim) po[1] = -y.y;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = t + L * q;
        const float2 y = yv[q];
        float* __restrict__ po = orow + o0 + 2 * n;
        if (2 * L * q < lim) {
          if (2 * n < lim) po[0] += y.x;
          if (2 * n + 1 < lim) po[1] -= y.y;
        }
      }
    }
    wave_sync();
  }
}

}  // namespace

extern "C" {

// Overlap-save FFT evaluation of at_fir_per_item_f32 (identical arguments and result up to
// rounding) -- the cost per output does not grow with the tap count.  `twiddles2048` is the
// device copy of at_stft_twiddles_host(2048, .).  Filters longer than 1536 taps are applied as
// partitions of 1024 taps, one launch each.  x and out must not alias.
int at_fir_fft_f32(const float* x, int64_t B, int64_t C, int64_t T, const float* taps, int taps_rows, int L_padded,
                   int half, int highpass, const float* twiddles2048, float* out, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !taps || !out || !twiddles2048 || x == out || B < 0 || C <= 0 || T <= 0 || T >= (1LL << 30) || L_padded <= 0 || half < 0 ||
      half >= L_padded || (taps_rows != 1 && taps_rows != B))
    return AT_ERR_INVALID;
  if (B == 0) return AT_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  FirFftArgs A;
  A.x = x; A.taps = taps; A.tw = reinterpret_cast<const float2*>(twiddles2048); A.out = out; A.T = T;
  A.rows = B * C; A.C = (int)C; A.taps_rows = taps_rows; A.Lp = L_padded; A.half = half; A.highpass = highpass;
  constexpr int MAX_SINGLE = 1536;  // longest filter handled as one partition (V = 513)
  constexpr int PART = 1024;        // partition length beyond that (V = 1024)
  const int part = L_padded <= MAX_SINGLE ? L_padded : PART;
  for (int j0 = 0; j0 < L_padded; j0 += part) {
    A.j0 = j0;
    A.nj = L_padded - j0 < part ? L_padded - j0 : part;
    A.accumulate = j0 > 0;
    A.V = (FF_N - A.nj + 1) & ~1;
    A.blocks_per_row = (int)((T + A.V - 1) / A.V);
    // enough (row, run) units to fill 256 CUs x 12 waves a few times over, runs of >= 8 blocks
    int64_t nseg = (16384 + A.rows - 1) / A.rows;
    const int max_seg = (A.blocks_per_row + 7) / 8;
    if (nseg > max_seg) nseg = max_seg;
    if (nseg < 1) nseg = 1;
    A.blocks_per_seg = (int)((A.blocks_per_row + nseg - 1) / nseg);
    A.nseg = (A.blocks_per_row + A.blocks_per_seg - 1) / A.blocks_per_seg;
    const int64_t units = A.rows * A.nseg;
    const int64_t wgs = (units + FF_NW - 1) / FF_NW;
    if (wgs > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fir_fft_kernel, dim3((unsigned)wgs), dim3(FF_NW * 64), 0, st, A);
    AT_LAUNCH_CHECK();
  }
  return AT_OK;
}

}  // extern "C"
