// Circular convolution of long signals by a four-step FFT, hand-written for gfx950 (no rocFFT).
//
// Replaces reference audiotools/core/effects.py:102-121 (EffectMixin.convolve):
//     rfft(ir, T), rfft(x, T), product, irfft(., T), scale by 1/clamp(max|ir|, 1e-5)
// -- a CIRCULAR convolution of length T (the reverb tail wraps to the head), kept as is.
//
// rocFFT needs 5 kernels per forward transform at T = 240000 (two transposes, two strided FFTs, the
// real post-processing), 13 launches and ~20 GB of traffic for one convolution (profiles/
// r02_cfg4_cfg5_kernel_stats.csv: 7 ms of the 12.8 ms cfg4 chain).  Here the real signal is read
// as M = T/2 complex points z[n] = x[2n] + i x[2n+1], i.e. as an N1 x N2 row-major matrix
// (n = N2 n1 + n2), and
//
//   colfft   FFT of every column (over n1): a 64-column tile sits in LDS, 512-byte segments in
//            and out.  Run on x (into `out`) and on the IR (into the workspace).
//   rowconv  one workgroup per row PAIR (k1, N1 - k1) of one signal: twiddle W_M^{k1 n2} and row
//            FFT (over n2) of x and of the IR in LDS -> the spectrum Z[k1 + N1 k2]; bins k and
//            M - k live in the two rows of the pair, so the real-FFT split, the product X H s / M,
//            the inverse split, the (conjugated) row FFT back and the twiddle all happen before
//            anything is written: the row pair is overwritten in place.
//   colfft   again on the result with a conjugated store: y.
//
// 4 launches, 4 reads + 3 writes of one signal-sized array (8.8 GB at cfg4 against 20 GB).
// The sub-FFTs are in-place mixed-radix (4, 2, 3, 5, 7) passes: every thread holds its butterflies
// in registers across the barrier, so a tile needs one LDS buffer, not two.
// T must be even with T/2 = N1 N2, N1 <= 512, N2 <= 2048, both {2,3,5,7}-smooth
// (at_longconv_supported); other lengths keep the rocFFT path (fftconv.hip).
#include "at_common.h"
#include "generic_fft.h"

#include <cmath>
#include <type_traits>
#include <cstring>

namespace {

using at::gfft::cmulf;
using at::gfft::dft_r;
using at::gfft::MAX_RADIX;
using at::gfft::PassList;
using at::gfft::TILE_POINTS;
using at::gfft::MAX_PASSES;
using at::gfft::ColLayout;
using at::gfft::RowLayout;
using at::gfft::run_passes;
using at::gfft::factor;

// A/B builds of the row kernel's organisation (AT_HIPCC_FLAGS): threads per row-pair workgroup and the longest row
#ifndef AT_ROW_THREADS
#define AT_ROW_THREADS 256
#endif
#ifndef AT_ROW_N2
#define AT_ROW_N2 2048
#endif
constexpr int THREADS = 256;
constexpr int MAX_N1 = 512;
constexpr int MAX_N2 = AT_ROW_N2;
struct Plan {
  int N1, N2;
  int cw, lcw;              // columns per colfft tile (power of two), log2
  int nhi;                  // ceil(N2 / 64): entries of the coarse row-twiddle table
  PassList p1, p2;
};

bool make_plan(int64_t T, Plan* P) {
  if (T < 2 || (T & 1) || T / 2 > (int64_t)MAX_N1 * MAX_N2) return false;
  const int64_t M = T / 2;
  {  // smooth?
    int64_t m = M;
    for (int r : {2, 3, 5, 7}) while (m % r == 0) m /= r;
    if (m != 1) return false;
  }
  // AT_LONGCONV_N2MAX (development aid): cap on the row length, to move work between the two kernels
  static const int n2_cap_env = at::env_int_once("AT_LONGCONV_N2MAX", MAX_N2);
  int64_t n2_cap = n2_cap_env;
  if (n2_cap < 1 || n2_cap > MAX_N2) n2_cap = MAX_N2;
  for (int64_t n2 = M < n2_cap ? M : n2_cap; n2 >= 1; --n2) {
    if (M % n2) continue;
    const int64_t n1 = M / n2;
    if (n1 > MAX_N1) return false;       // n1 only grows from here
    P->N1 = (int)n1; P->N2 = (int)n2;
    if (!factor(P->N1, &P->p1) || !factor(P->N2, &P->p2)) continue;
    int cw = 64;
    while (cw > 1 && P->N1 * cw > TILE_POINTS) cw >>= 1;
    P->cw = cw; P->lcw = 0;
    while ((1 << P->lcw) < cw) ++P->lcw;
    P->nhi = (P->N2 + 63) / 64;
    return true;
  }
  return false;
}

// table layout (float2 units)
struct TableOffsets { int64_t tw1, tw2, rowtw, sp_lo, sp_hi, total; int rt; };
TableOffsets table_offsets(const Plan& P) {
  TableOffsets o;
  o.rt = 64 + P.nhi;
  o.tw1 = 0;
  o.tw2 = o.tw1 + P.N1;
  o.rowtw = o.tw2 + P.N2;
  o.sp_lo = o.rowtw + (int64_t)P.N1 * o.rt;
  o.sp_hi = o.sp_lo + P.N1;
  o.total = o.sp_hi + P.N2;
  return o;
}

// ---------------------------------------------------------------- column FFT
struct ColArgs {
  const float2* src;        // (rows, N1, N2)
  float2* dst;              // (rows, N1, N2); may alias src (a workgroup owns its columns)
  const float2* tw1;        // w_N1^t
  int64_t rows;
  int N1, N2, cw, lcw, tiles;
  // ROLLED source (the impulse response as the caller holds it): real rows of `src_pitch` floats
  // whose first `src_len` samples are valid and the rest of the period T = 2 N1 N2 is zero, read
  // rotated by shift[row] (effects.py:94-100: roll so that the peak sits at sample 0)
  const float* rsrc;
  const int64_t* shift;     // (rows) or null
  int64_t src_pitch, src_len;
  unsigned* peak;           // (rows) max |sample| as float bits (PEAK variants), zeroed by the caller
  PassList pl;
};

constexpr int COL_CONJ = 1, COL_ROLLED = 2, COL_PEAK = 4;

// max over the wave, then one atomic per wave.  |x| of a float compares like its bit pattern, and a
// NaN (>= 0x7f800001) wins the integer max: the peak propagates NaN as at_absmax_f32 does.
__device__ __forceinline__ void peak_commit(float m, unsigned* slot) {
  unsigned u = __float_as_uint(m) & 0x7fffffffu;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)u, d, 64);
    u = o > u ? o : u;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(slot, u);
}

__device__ __forceinline__ unsigned absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

template <int MODE>
__global__ __launch_bounds__(THREADS, 4) void colfft_kernel(const ColArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 smem[];
  float2* tile = smem;                       // [N1][cw]
  float2* tw = smem + (A.N1 << A.lcw);       // [N1]
  const int64_t row = blockIdx.x / A.tiles;
  const int t = (int)(blockIdx.x - row * A.tiles);
  const int n2_0 = t << A.lcw;
  const int cmask = A.cw - 1;
  const int npts = A.N1 << A.lcw;
  float2* __restrict__ dst = A.dst + row * (int64_t)A.N1 * A.N2;
  unsigned pk = 0;
  at::gfft::build_pass_twiddles<THREADS>(tw, A.tw1, 1, A.N1, A.pl);     // per-pass blocks (conflict-free reads)
  // The tile's points are loaded by a FIXED number of unconditional, address-clamped loads per thread into registers and
  // only then written to LDS: with a predicated loop every iteration was its own basic block (load, wait, ds_write), i.e.
  // ONE 512-byte segment in flight per wave -- 8 KB per CU where HBM latency x bandwidth needs ~60 (the kernel ran at
  // 39 % of 8 TB/s with 72 % of its wave cycles waiting, profiles/r03_notes.md).
  constexpr int CPT = TILE_POINTS / THREADS;       // points per thread
  // point e = tid + THREADS i of the tile is (row n1_0 + step i, column col): cw divides THREADS
  const int step = THREADS >> A.lcw, n1_0 = (int)threadIdx.x >> A.lcw;
  const int n2 = n2_0 + ((int)threadIdx.x & cmask);
  const bool inside = n2 < A.N2;
  const int n2c = inside ? n2 : A.N2 - 1;
  float2 r[CPT];
  if constexpr (MODE & COL_ROLLED) {
    const float* __restrict__ rs = A.rsrc + row * A.src_pitch;
    const int64_t T = 2 * (int64_t)A.N1 * A.N2;
    int64_t sh = A.shift ? A.shift[row] % T : 0;
    if (sh < 0) sh += T;
    const int64_t last = A.src_len > 0 ? A.src_len - 1 : 0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int n1 = min(n1_0 + step * i, A.N1 - 1);
      int64_t i0 = 2 * (int64_t)(__mul24(n1, A.N2) + n2c) + sh;
      if (i0 >= T) i0 -= T;
      int64_t i1 = i0 + 1;
      if (i1 >= T) i1 -= T;
      const float a = rs[i0 < last ? i0 : last], b = rs[i1 < last ? i1 : last];
      r[i] = make_float2(i0 < A.src_len ? a : 0.f, i1 < A.src_len ? b : 0.f);
    }
  } else {
    const float2* __restrict__ src = A.src + row * (int64_t)A.N1 * A.N2;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int n1 = min(n1_0 + step * i, A.N1 - 1);
      r[i] = src[__mul24(n1, A.N2) + n2c];        // (32-bit offset from the row: N1 N2 <= 2^20 points)
    }
  }
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    if (n1_0 + step * i < A.N1) {
      const float2 v = inside ? r[i] : make_float2(0.f, 0.f);
      if constexpr ((MODE & COL_PEAK) && !(MODE & COL_CONJ)) { pk = max(pk, max(absbits(v.x), absbits(v.y))); }
      tile[(int)threadIdx.x + THREADS * i] = v;
    }
  }
  __syncthreads();
  run_passes<THREADS>(tile, tw, A.N1, A.pl, A.cw, ColLayout{A.lcw, cmask});
  // results: LDS -> registers (all reads in flight), then the guarded stores
#pragma unroll
  for (int i = 0; i < CPT; ++i) r[i] = tile[min((int)threadIdx.x + THREADS * i, npts - 1)];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int k1 = n1_0 + step * i;
    if (k1 < A.N1 && inside) {
      float2 v = r[i];
      if constexpr (MODE & COL_CONJ) v.y = -v.y;
      if constexpr ((MODE & COL_PEAK) && (MODE & COL_CONJ)) { pk = max(pk, max(absbits(v.x), absbits(v.y))); }
      dst[__mul24(k1, A.N2) + n2] = v;
    }
  }
  if constexpr (MODE & COL_PEAK) peak_commit(__uint_as_float(pk), A.peak + row);
}

// ---------------------------------------------------------------- row FFTs + spectrum product
struct RowArgs {
  float2* ax;               // (B*C, N1, N2): column-transformed signal, overwritten with the result
  const float2* ah;         // (B*Cir, N1, N2): column-transformed impulse response
  const float* scale;       // (B*Cir) or null
  const float2* tw2;        // w_N2^t
  const float2* rowtw;      // (N1, 64 + nhi): w_M^{k1 b}, b < 64; w_M^{64 k1 a}, a < nhi
  const float2* sp_lo;      // w_T^{k1}
  const float2* sp_hi;      // w_T^{N1 k2}
  int C, Cir;
  int N1, N2, rt, npairs;
  float inv_m;
  int64_t total_units;      // rows * npairs (the wave form: one wave per unit)
  PassList pl;
};

// Threads per row-pair workgroup and workgroups per CU.  Measured at cfg4 (rowconv_kernel alone):
// 256 x 2 (176 registers) 2.7 ms, 256 x 3 (168 registers, 8 dwords spilled) 2.0 ms, 512 x 2 (98
// registers) 2.0 ms, 1024 x 2 (59 registers) 2.2 ms: past three independent workgroups per CU the
// wave organisation does not matter (profiles/r02_notes.md).
constexpr int RTHREADS = AT_ROW_THREADS;
constexpr int ROW_WGS = 3 * 256 / RTHREADS;           // three waves per SIMD
constexpr int ROW_LOADS = 2 * MAX_N2 / RTHREADS;      // points of a row pair per thread
constexpr int SPEC_ITERS = MAX_N2 / RTHREADS;         // bins of one row per thread

// X[k], X[M-k] of the real signal from Z[k], Z[M-k] of its half-length complex transform; w = w_T^k.
__device__ __forceinline__ void real_split(float2 zk, float2 zm, float2 w, float2& xk, float2& xm) {
  // X[k] = (zk + conj zm)/2 - i w (zk - conj zm)/2;  X[M-k] = conj(the same with +)
  const float sr = zk.x + zm.x, si = zk.y - zm.y;      // zk + conj zm
  const float dr = zk.x - zm.x, di = zk.y + zm.y;      // zk - conj zm
  // -i w d = -i (wx + i wy)(dr + i di) = (wx di + wy dr) + i (wy di - wx dr)
  const float pr = fmaf(w.x, di, w.y * dr), pi = fmaf(w.y, di, -w.x * dr);
  xk = make_float2(0.5f * (sr + pr), 0.5f * (si + pi));
  xm = make_float2(0.5f * (sr - pr), 0.5f * (-si + pi));
}

// the inverse: Z'[k], Z'[M-k] of the half-length inverse transform from Y[k], Y[M-k]
__device__ __forceinline__ void real_merge(float2 yk, float2 ym, float2 w, float2& zk, float2& zm) {
  // Z'[k] = (yk + conj ym)/2 + i conj(w) (yk - conj ym)/2;  Z'[M-k] = conj of the same with -
  const float sr = yk.x + ym.x, si = yk.y - ym.y;
  const float dr = yk.x - ym.x, di = yk.y + ym.y;
  // i conj(w) d = i (wx - i wy)(dr + i di) = (wy dr - wx di) + i (wx dr + wy di)
  const float pr = fmaf(w.y, dr, -w.x * di), pi = fmaf(w.x, dr, w.y * di);
  zk = make_float2(0.5f * (sr + pr), 0.5f * (si + pi));
  zm = make_float2(0.5f * (sr - pr), 0.5f * (-si + pi));
}

// LDS: 49.5 KB per workgroup at N2 = 2000.  VALU (~0.75 ms of issue), LDS (~0.7 ms) and HBM (~0.75 ms)
// time add up rather than overlap: the four waves of a workgroup move in lockstep between barriers
// and only three workgroups share a CU.
__global__ __launch_bounds__(RTHREADS, ROW_WGS * RTHREADS / 256) void rowconv_kernel(const RowArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 smem[];
  const int N2 = A.N2;
  float2* buf = smem;                       // [2][N2]
  float2* tw = smem + 2 * N2;               // [N2]
  float2* rt = tw + N2;                     // [2][rt]
  const int64_t xrow = blockIdx.x / A.npairs;
  const int p = (int)(blockIdx.x - xrow * A.npairs);
  const int k1a = p, k1b = (A.N1 - p) % A.N1;
  const bool self = k1a == k1b;
  const int nrow = self ? 1 : 2;
  const int64_t b = xrow / A.C;
  const int c = (int)(xrow - b * A.C);
  const int64_t hrow = b * A.Cir + (A.Cir == 1 ? 0 : c);
  const int64_t MM = (int64_t)A.N1 * N2;
  float2* __restrict__ gx = A.ax + xrow * MM;
  const float2* __restrict__ gh = A.ah + hrow * MM;
  const int npts = nrow * N2;

  const RowLayout lay{N2, 0};
  const float2 wlo = A.sp_lo[k1a];
  const int slot_b = self ? 0 : N2;
  const float sc = (A.scale ? A.scale[hrow] : 1.0f) * A.inv_m;

  // The load / twiddle / store loops have a compile-time trip count (NI = 16 points per thread) and clamped indices
  // instead of a run-time "is this iteration inside the row pair" test per iteration: that test (uniform) made every
  // iteration its own basic block, so the LDS reads of the row twiddles and the multiplication of one point had to
  // finish before the next point's reads were issued.  Surplus iterations (N2 < 2048, the two self-paired rows)
  // repeat the last point: same value to the same slot / address.
  auto body = [&](auto ni_c) __attribute__((always_inline)) {
    constexpr int NI = decltype(ni_c)::value;
    float2 r[NI];
    auto fetch = [&](const float2* __restrict__ g) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int e = min((int)threadIdx.x + RTHREADS * i, npts - 1);
        const int s = e >= N2 ? 1 : 0;
        r[i] = g[__mul24(s ? k1b : k1a, N2) + (e - s * N2)];
      }
    };
    fetch(gx);
    at::gfft::build_pass_twiddles<RTHREADS>(tw, A.tw2, 1, N2, A.pl);       // per-pass blocks (conflict-free reads)
    for (int i = threadIdx.x; i < nrow * A.rt; i += RTHREADS) {
      const int s = i >= A.rt ? 1 : 0;
      rt[i] = A.rowtw[(int64_t)(s ? k1b : k1a) * A.rt + (i - s * A.rt)];
    }
    __syncthreads();
    float2 Xk[SPEC_ITERS], Xm[SPEC_ITERS];

    // phase 0: signal rows -> X;  phase 1: IR rows -> H, Y = X H, Z' (conjugated) back into the
    // slots;  phase 2: the transform back.  One copy of the pass code serves all three.
    for (int ph = 0; ph < 3; ++ph) {
      int tid = (int)threadIdx.x;             // opaque per phase: nothing below is hoisted out of the loop
      asm volatile("" : "+v"(tid));
      if (ph < 2) {                           // twiddle w_M^{k1 n2} and into LDS
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int e = min(tid + RTHREADS * i, npts - 1);
          const int s = e >= N2 ? 1 : 0;
          const int n2 = e - s * N2;
          const float2 w = cmulf(rt[s * A.rt + (n2 & 63)], rt[s * A.rt + 64 + (n2 >> 6)]);
          buf[e] = cmulf(r[i], w);
          if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // batches of four: all sixteen at once spill
        }
        __syncthreads();
        if (ph == 0) fetch(gh);               // the IR rows arrive while the signal rows are transformed
      }
      run_passes<RTHREADS>(buf, tw, N2, A.pl, nrow, lay);
      if (ph == 2) break;
      // Every bin pair (k, M - k) is read (and in phase 1 rewritten) by exactly one thread, so the
      // product spectrum goes straight back into the slots it came from.  (Staging the reads of this loop -- two slots
      // and the L2-resident split twiddle per pair -- ahead of the arithmetic in groups of four measured 2.5 % slower,
      // 2.69 -> 2.76 ms for the whole convolution: 20 more spilled registers.)  The inverse transform
      // runs as a forward one on conj(Z'); colfft<true> conjugates at the end.
#pragma unroll
      for (int i = 0; i < SPEC_ITERS; ++i) {
        const int k2 = tid + RTHREADS * i;
        const int k2m = k1a == 0 ? (k2 == 0 ? 0 : N2 - k2) : N2 - 1 - k2;
        if (k2 < N2 && (!self || k2 <= k2m)) {
          const float2 zk = buf[k2], zm = buf[slot_b + k2m];
          const bool dc = k1a == 0 && k2 == 0;            // DC and Nyquist, both real, packed in Z[0]
          const float2 w = cmulf(wlo, A.sp_hi[k2]);
          float2 sk, sm;
          if (dc) { sk = make_float2(zk.x + zk.y, 0.f); sm = make_float2(zk.x - zk.y, 0.f); }
          else real_split(zk, zm, w, sk, sm);
          if (ph == 0) { Xk[i] = sk; Xm[i] = sm; }
          else {
            float2 ok, om;
            if (dc) {
              const float y0 = Xk[i].x * sk.x * sc, ym = Xm[i].x * sm.x * sc;
              ok = make_float2(0.5f * (y0 + ym), 0.5f * (y0 - ym));
              om = ok;
            } else {
              float2 yk = cmulf(Xk[i], sk), ym = cmulf(Xm[i], sm);
              yk.x *= sc; yk.y *= sc; ym.x *= sc; ym.y *= sc;
              real_merge(yk, ym, w, ok, om);
            }
            buf[k2] = make_float2(ok.x, -ok.y);
            if (!(self && k2 == k2m)) buf[slot_b + k2m] = make_float2(om.x, -om.y);
          }
        }
      }
      __syncthreads();
    }
    int tid2 = (int)threadIdx.x;              // opaque: the store addresses equal the first fetch's, and the compiler
    asm volatile("" : "+v"(tid2));            // kept all sixteen alive (spilled) across the whole kernel
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = min(tid2 + RTHREADS * i, npts - 1);
      const int s = e >= N2 ? 1 : 0;
      const int n2 = e - s * N2;
      const float2 w = cmulf(rt[s * A.rt + (n2 & 63)], rt[s * A.rt + 64 + (n2 >> 6)]);
      gx[__mul24(s ? k1b : k1a, N2) + n2] = cmulf(buf[e], w);
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };
  body(std::integral_constant<int, ROW_LOADS>{});
}

#if AT_DEV_KNOBS
// ---------------------------------------------------------------- the same, ONE WAVE per row pair (round 5; development build)
// MEASURED, NOT SHIPPED (profiles/r05_notes.md section 3): at cfg4 on the N2 <= 1024 plan (N1 = 120, N2 = 1000) this kernel
// takes 1.49 ms where rowconv_kernel takes 1.40 ms on the N2 = 2000 plan; with the column kernels 10 % faster at N1 = 120 the
// whole convolution is 2.66-2.69 ms against 2.69-2.70 ms: a draw.  Counters: 9 720 VALU instructions per row pair and lane
// (104 per complex point and transform), VALU active 42 % of the wave cycles x 2 waves per SIMD -- without the barriers the
// row kernel is bound by the instruction count of the generic mixed-radix passes, not by latency any more.
// rowconv_kernel's four waves move in lockstep between ~28 workgroup barriers and three workgroups share a CU: VALU, LDS and
// HBM time add up (1.40 ms for 3.02 GB at cfg4 = 27 % of 8 TB/s, 39 % issue utilisation).  Here a row pair belongs to one
// wave: its two rows live in the wave's own LDS slab, the mixed-radix passes are the same code with the lane as thread
// index and a wave-level fence where the workgroup met (generic_fft.h, WAVE = true), so no wave ever waits for another one
// and eight row pairs per CU are in eight different phases -- loads, passes, split step, stores overlap across waves instead
// of inside one.  Rows of at most 1024 points (2 x N2 x 8 B per wave: two workgroups of four waves per CU); the arithmetic,
// tables and index algebra are rowconv_kernel's, statement for statement.
constexpr int WROW_N2 = 1024;                   // longest row of the wave form
constexpr int WROW_WAVES = 4;                   // waves (row pairs) per workgroup
constexpr int WLOADS = 2 * WROW_N2 / 64;        // points of a row pair per lane
constexpr int WSPEC = WROW_N2 / 64;             // bins of one row per lane

// PLAN = 0: run-time pass list (any eligible row length); PLAN = 1: N2 = 1000 = 25 * 5 * 8 with the pass list as literals
// (cfg4's T = 240000 on the N2 <= 1024 plan): index arithmetic folds, and the kernel is 1/8 of the generic one's 128 KB of
// code -- eight waves per CU in eight different phases have to share the instruction cache.
template <int PLAN>
__global__ __launch_bounds__(WROW_WAVES * 64, 2) void rowconv_wave_kernel(const RowArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 smem[];
  const int N2 = PLAN == 1 ? 1000 : A.N2;
  float2* tw = smem;                                            // [N2] per-pass twiddle blocks, shared by the workgroup
  at::gfft::build_pass_twiddles<WROW_WAVES * 64>(tw, A.tw2, 1, N2, A.pl);
  __syncthreads();                                              // the only workgroup barrier
  const int lane = (int)(threadIdx.x & 63);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float2* buf = smem + N2 + wave * (2 * N2 + 2 * A.rt);         // [2][N2]
  float2* rt = buf + 2 * N2;                                    // [2][rt]
  const int64_t unit = (int64_t)blockIdx.x * WROW_WAVES + wave;
  if (unit >= A.total_units) return;                            // wave-uniform
  const int64_t xrow = unit / A.npairs;
  const int p = (int)(unit - xrow * A.npairs);
  const int k1a = p, k1b = (A.N1 - p) % A.N1;
  const bool self = k1a == k1b;
  const int nrow = self ? 1 : 2;
  const int64_t b = xrow / A.C;
  const int c = (int)(xrow - b * A.C);
  const int64_t hrow = b * A.Cir + (A.Cir == 1 ? 0 : c);
  const int64_t MM = (int64_t)A.N1 * N2;
  float2* __restrict__ gx = A.ax + xrow * MM;
  const float2* __restrict__ gh = A.ah + hrow * MM;
  const int npts = nrow * N2;
  const RowLayout lay{N2, 0};
  const float2 wlo = A.sp_lo[k1a];
  const int slot_b = self ? 0 : N2;
  const float sc = (A.scale ? A.scale[hrow] : 1.0f) * A.inv_m;

  for (int i = lane; i < nrow * A.rt; i += 64) {
    const int s = i >= A.rt ? 1 : 0;
    rt[i] = A.rowtw[(int64_t)(s ? k1b : k1a) * A.rt + (i - s * A.rt)];
  }
  at::wave_sync();

  // rows of one signal -> twiddle w_M^{k1 n2} -> the wave's slab.  All loads of the pair are issued before the first
  // product (fixed trip count, clamped indices: surplus iterations repeat the last point).
  auto load_rows = [&](const float2* __restrict__ g) __attribute__((always_inline)) {
    float2 r[WLOADS];
    int tid = lane;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int e = min(tid + 64 * i, npts - 1);
      const int s = e >= N2 ? 1 : 0;
      r[i] = g[__mul24(s ? k1b : k1a, N2) + (e - s * N2)];
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int e = min(tid + 64 * i, npts - 1);
      const int s = e >= N2 ? 1 : 0;
      const int n2 = e - s * N2;
      const float2 w = cmulf(rt[s * A.rt + (n2 & 63)], rt[s * A.rt + 64 + (n2 >> 6)]);
      buf[e] = cmulf(r[i], w);
      if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    at::wave_sync();
  };

  float2 Xk[WSPEC], Xm[WSPEC];
  for (int ph = 0; ph < 3; ++ph) {
    int tid = lane;                           // opaque per phase: nothing below is hoisted out of the loop
    asm volatile("" : "+v"(tid));
    if (ph < 2) load_rows(ph == 0 ? reinterpret_cast<const float2*>(gx) : gh);
    if constexpr (PLAN == 1) {
      // butterflies per lane for a row pair: 80 / 64 -> 2, 400 / 64 -> 7, 250 / 64 -> 4 (a self-paired row: half of them,
      // the surplus lanes repeat the last butterfly); twiddle blocks: radix 5 behind 25 at 0, radix 8 behind 125 at 100
      at::gfft::pass_inplace<25, 2, 64, RowLayout, true>(buf, tw, 1000, 1, nrow * 40, lay);
      at::gfft::pass_inplace<5, 7, 64, RowLayout, true>(buf, tw, 1000, 25, nrow * 200, lay);
      at::gfft::pass_inplace<8, 4, 64, RowLayout, true>(buf, tw + 100, 1000, 125, nrow * 125, lay);
    } else {
      run_passes<64, RowLayout, true>(buf, tw, N2, A.pl, nrow, lay);
    }
    if (ph == 2) break;
#pragma unroll
    for (int i = 0; i < WSPEC; ++i) {
      const int k2 = tid + 64 * i;
      const int k2m = k1a == 0 ? (k2 == 0 ? 0 : N2 - k2) : N2 - 1 - k2;
      if (k2 < N2 && (!self || k2 <= k2m)) {
        const float2 zk = buf[k2], zm = buf[slot_b + k2m];
        const bool dc = k1a == 0 && k2 == 0;            // DC and Nyquist, both real, packed in Z[0]
        const float2 w = cmulf(wlo, A.sp_hi[k2]);
        float2 sk, sm;
        if (dc) { sk = make_float2(zk.x + zk.y, 0.f); sm = make_float2(zk.x - zk.y, 0.f); }
        else real_split(zk, zm, w, sk, sm);
        if (ph == 0) { Xk[i] = sk; Xm[i] = sm; }
        else {
          float2 ok, om;
          if (dc) {
            const float y0 = Xk[i].x * sk.x * sc, ym = Xm[i].x * sm.x * sc;
            ok = make_float2(0.5f * (y0 + ym), 0.5f * (y0 - ym));
            om = ok;
          } else {
            float2 yk = cmulf(Xk[i], sk), ym = cmulf(Xm[i], sm);
            yk.x *= sc; yk.y *= sc; ym.x *= sc; ym.y *= sc;
            real_merge(yk, ym, w, ok, om);
          }
          buf[k2] = make_float2(ok.x, -ok.y);
          if (!(self && k2 == k2m)) buf[slot_b + k2m] = make_float2(om.x, -om.y);
        }
      }
    }
    at::wave_sync();
  }
  int tid2 = lane;
  asm volatile("" : "+v"(tid2));
#pragma unroll
  for (int i = 0; i < WLOADS; ++i) {
    const int e = min(tid2 + 64 * i, npts - 1);
    const int s = e >= N2 ? 1 : 0;
    const int n2 = e - s * N2;
    const float2 w = cmulf(rt[s * A.rt + (n2 & 63)], rt[s * A.rt + 64 + (n2 >> 6)]);
    gx[__mul24(s ? k1b : k1a, N2) + n2] = cmulf(buf[e], w);
    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
}

#endif  // AT_DEV_KNOBS

inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

}  // namespace

extern "C" {

// 1 when T has a four-step plan (even, T/2 = N1 N2 with N1 <= 512, N2 <= 2048, {2,3,5,7}-smooth)
int at_longconv_supported(int64_t T) {
  Plan P;
  return make_plan(T, &P) ? 1 : 0;
}

// the split chosen for T (for tests and documentation)
int at_longconv_plan(int64_t T, int* n1, int* n2) {
  Plan P;
  if (!make_plan(T, &P)) return AT_ERR_UNSUPPORTED;
  if (n1) *n1 = P.N1;
  if (n2) *n2 = P.N2;
  return AT_OK;
}

// floats of the twiddle tables of length T
int64_t at_longconv_table_floats(int64_t T) {
  Plan P;
  if (!make_plan(T, &P)) return AT_ERR_UNSUPPORTED;
  return 2 * table_offsets(P).total;
}

// fills the tables (host memory, `n` floats as at_longconv_table_floats says), evaluated in double
int at_longconv_tables_host(int64_t T, float* out, int64_t n) {
  Plan P;
  if (!out || !make_plan(T, &P)) return out ? AT_ERR_UNSUPPORTED : AT_ERR_INVALID;
  const TableOffsets o = table_offsets(P);
  if (n != 2 * o.total) return AT_ERR_INVALID;
  const int64_t M = T / 2;
  auto put = [&](int64_t at, int64_t num, int64_t den) {    // (cos, -sin)(2 pi num / den)
    num %= den;
    const double a = 2.0 * M_PI * (double)num / (double)den;
    out[2 * at] = (float)std::cos(a);
    out[2 * at + 1] = (float)(-std::sin(a));
  };
  for (int t = 0; t < P.N1; ++t) put(o.tw1 + t, t, P.N1);
  for (int t = 0; t < P.N2; ++t) put(o.tw2 + t, t, P.N2);
  for (int k1 = 0; k1 < P.N1; ++k1) {
    for (int b = 0; b < 64; ++b) put(o.rowtw + (int64_t)k1 * o.rt + b, (int64_t)k1 * b, M);
    for (int a = 0; a < P.nhi; ++a) put(o.rowtw + (int64_t)k1 * o.rt + 64 + a, (int64_t)k1 * 64 * a, M);
  }
  for (int k1 = 0; k1 < P.N1; ++k1) put(o.sp_lo + k1, k1, T);
  for (int k2 = 0; k2 < P.N2; ++k2) put(o.sp_hi + k2, (int64_t)P.N1 * k2, T);
  return AT_OK;
}

// scratch: the column-transformed impulse responses
int64_t at_longconv_workspace_bytes(int64_t B, int64_t C, int64_t Cir, int64_t T) {
  if (B < 0 || C <= 0 || Cir <= 0 || T <= 0) return AT_ERR_INVALID;
  return align256(B * Cir * T * 4);
}

}  // extern "C"

namespace {

struct IrSource {           // how the impulse response is held by the caller
  const float* ptr;         // (B*Cir, pitch)
  int64_t pitch, len;       // row pitch and valid prefix (samples at and after `len` count as zero)
  const int64_t* shift;     // (B*Cir) rotation to apply while reading, or null
  bool rolled;              // false: plain (B*Cir, T) rows, read as float2
};

int run_longconv(const float* x, const IrSource& ir, const float* scale, int64_t B, int64_t C, int64_t Cir, int64_t T,
                 const float* tables, float* out, float* x_peak, float* y_peak, void* workspace, int64_t workspace_bytes,
                 hipStream_t st) {
  Plan P;
  if (!make_plan(T, &P)) return AT_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < at_longconv_workspace_bytes(B, C, Cir, T)) return AT_ERR_INVALID;
  const TableOffsets o = table_offsets(P);
  const float2* tb = reinterpret_cast<const float2*>(tables);
  const int tiles = (P.N2 + P.cw - 1) / P.cw;
  const int64_t rows_x = B * C, rows_h = B * Cir;
  if (rows_x * tiles > 0x7fffffffLL || rows_x * (P.N1 / 2 + 1) > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;

  ColArgs ca;
  ca.tw1 = tb + o.tw1; ca.N1 = P.N1; ca.N2 = P.N2; ca.cw = P.cw; ca.lcw = P.lcw; ca.tiles = tiles; ca.pl = P.p1;
  ca.rsrc = nullptr; ca.shift = nullptr; ca.src_pitch = 0; ca.src_len = 0; ca.peak = nullptr;
  const size_t col_lds = ((size_t)P.N1 * P.cw + P.N1) * sizeof(float2);
  const size_t row_lds = ((size_t)3 * P.N2 + 2 * o.rt) * sizeof(float2);
  int e;
  if ((e = at::allow_big_lds(reinterpret_cast<const void*>(rowconv_kernel))) != AT_OK) return e;
  for (float* pkbuf : {x_peak, y_peak}) {
    if (!pkbuf) continue;
    const hipError_t he = hipMemsetAsync(pkbuf, 0, rows_x * sizeof(float), st);
    if (he != hipSuccess) return AT_ERR_HIP(he);
  }

  float2* ah = reinterpret_cast<float2*>(workspace);
  float2* ax = reinterpret_cast<float2*>(out);
  const dim3 grid_h((unsigned)(rows_h * tiles)), grid_x((unsigned)(rows_x * tiles));
  ca.dst = ah; ca.rows = rows_h;
  if (ir.rolled) {
    ca.src = nullptr; ca.rsrc = ir.ptr; ca.shift = ir.shift; ca.src_pitch = ir.pitch; ca.src_len = ir.len;
    hipLaunchKernelGGL(colfft_kernel<COL_ROLLED>, grid_h, dim3(THREADS), col_lds, st, ca);
  } else {
    ca.src = reinterpret_cast<const float2*>(ir.ptr);
    hipLaunchKernelGGL(colfft_kernel<0>, grid_h, dim3(THREADS), col_lds, st, ca);
  }
  AT_LAUNCH_CHECK();
  ca.src = reinterpret_cast<const float2*>(x); ca.dst = ax; ca.rows = rows_x;
  ca.peak = reinterpret_cast<unsigned*>(x_peak);
  if (x_peak) hipLaunchKernelGGL(colfft_kernel<COL_PEAK>, grid_x, dim3(THREADS), col_lds, st, ca);
  else hipLaunchKernelGGL(colfft_kernel<0>, grid_x, dim3(THREADS), col_lds, st, ca);
  AT_LAUNCH_CHECK();

  RowArgs ra;
  ra.ax = ax; ra.ah = ah; ra.scale = scale; ra.tw2 = tb + o.tw2; ra.rowtw = tb + o.rowtw; ra.sp_lo = tb + o.sp_lo;
  ra.sp_hi = tb + o.sp_hi; ra.C = (int)C; ra.Cir = (int)Cir; ra.N1 = P.N1; ra.N2 = P.N2; ra.rt = o.rt;
  ra.npairs = P.N1 / 2 + 1; ra.inv_m = 1.0f / (float)(T / 2); ra.pl = P.p2;
  ra.total_units = rows_x * ra.npairs;
#if AT_DEV_KNOBS
  static const int wave_form = at::env_int_once("AT_ROWCONV_WAVE", 0);      // development A/B: 1 = one wave per row pair (rows <= 1024)
  if (wave_form && P.N2 <= WROW_N2 && at::gfft::wave_pass_list_ok(P.p2, P.N2, 2)) {
    const size_t wlds = ((size_t)P.N2 + (size_t)WROW_WAVES * (2 * P.N2 + 2 * o.rt)) * sizeof(float2);
    const bool fixed1 = P.N2 == 1000 && P.p2.n == 3 && P.p2.radix[0] == 25 && P.p2.radix[1] == 5 && P.p2.radix[2] == 8;
    const void* kfn = fixed1 ? reinterpret_cast<const void*>(rowconv_wave_kernel<1>) : reinterpret_cast<const void*>(rowconv_wave_kernel<0>);
    if ((e = at::allow_big_lds(kfn)) != AT_OK) return e;
    const int64_t wgs = (ra.total_units + WROW_WAVES - 1) / WROW_WAVES;
    if (fixed1) hipLaunchKernelGGL(rowconv_wave_kernel<1>, dim3((unsigned)wgs), dim3(WROW_WAVES * 64), wlds, st, ra);
    else hipLaunchKernelGGL(rowconv_wave_kernel<0>, dim3((unsigned)wgs), dim3(WROW_WAVES * 64), wlds, st, ra);
  } else
#endif
  {
    hipLaunchKernelGGL(rowconv_kernel, dim3((unsigned)(rows_x * ra.npairs)), dim3(RTHREADS), row_lds, st, ra);
  }
  AT_LAUNCH_CHECK();

  ca.src = ax; ca.dst = ax; ca.rows = rows_x;
  ca.peak = reinterpret_cast<unsigned*>(y_peak);
  if (y_peak) hipLaunchKernelGGL(colfft_kernel<COL_CONJ | COL_PEAK>, grid_x, dim3(THREADS), col_lds, st, ca);
  else hipLaunchKernelGGL(colfft_kernel<COL_CONJ>, grid_x, dim3(THREADS), col_lds, st, ca);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // namespace

extern "C" {

// x (B,C,T), ir (B,Cir,T) with Cir == 1 or Cir == C, scale (B,Cir) or NULL, tables (device copy of
// at_longconv_tables_host), out (B,C,T): out = irfft(rfft(x) rfft(ir)) * scale.
int at_longconv_circ_f32(const float* x, const float* ir, const float* scale, int64_t B, int64_t C, int64_t Cir, int64_t T,
                         const float* tables, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (B == 0) return AT_OK;
  if (!x || !ir || !out || !tables || B < 0 || C <= 0 || T <= 0 || (Cir != 1 && Cir != C)) return AT_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ir) | reinterpret_cast<uintptr_t>(out)) & 7)
    return AT_ERR_INVALID;
  const IrSource src{ir, T, T, nullptr, false};
  return run_longconv(x, src, scale, B, C, Cir, T, tables, out, nullptr, nullptr, workspace, workspace_bytes,
                      reinterpret_cast<hipStream_t>(stream));
}

// The convolution as EffectMixin.apply_ir needs it (effects.py:92-121 with :160 and :175 folded in):
//   ir (B*Cir, ir_pitch): the impulse responses as they are, `ir_len` <= T valid samples per row (the
//     zero padding to T of effects.py:86-90 is implied, not read) and ir_shift (B*Cir) int64 or NULL:
//     the rotation that puts the peak at sample 0 (effects.py:94-100), applied while reading;
//   x_peak, y_peak (B*C) or NULL: max |x| and max |out| per row, found while the column transforms
//     stream x in and the result out (NaN propagates, as in at_absmax_f32).
int at_longconv_room_f32(const float* x, const float* ir, int64_t ir_pitch, int64_t ir_len, const int64_t* ir_shift,
                         const float* scale, int64_t B, int64_t C, int64_t Cir, int64_t T, const float* tables, float* out,
                         float* x_peak, float* y_peak, void* workspace, int64_t workspace_bytes, void* stream) {
  if (B == 0) return AT_OK;
  if (!x || !ir || !out || !tables || B < 0 || C <= 0 || T <= 0 || (Cir != 1 && Cir != C) || ir_len < 0 || ir_len > T ||
      ir_pitch < ir_len)
    return AT_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 7) return AT_ERR_INVALID;
  const IrSource src{ir, ir_pitch, ir_len, ir_shift, true};
  return run_longconv(x, src, scale, B, C, Cir, T, tables, out, x_peak, y_peak, workspace, workspace_bytes,
                      reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
