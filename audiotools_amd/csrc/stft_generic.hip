// Generic-size STFT / inverse-STFT frames for gfx950: n_fft even, n_fft / 2 = 2^a 3^b 5^c 7^d, up to 16384.
//
// Covers what the wave-FFT kernels (csrc/stft.hip, csrc/istft.hip: powers of two up to 2048) do
// not: the reference's default window for 96 / 192 kHz audio (audio_signal.py:1066-1070 gives
// 4096 / 8192) and speech front ends with 400- / 1200- / 1920-sample windows.  Without this the
// package fell back to torch.stft (hipFFT on a materialised frame matrix) for those sizes.
//
// One workgroup transforms one frame at a time.  The real FFT of length N is a complex FFT of
// length M = N / 2 on z[n] = x[2n] + i x[2n+1] plus the split step (as in stft.hip); the complex
// FFT is a Stockham autosort with mixed radices 4 / 2 / 3 / 5 / 7 between two LDS buffers, one
// __syncthreads() per pass.  Twiddles come from the (cos, -sin)(2 pi k / N) table the fast
// kernels use (w_M^k = table[2 k]).  It is a correct, HBM-class path (LDS traffic per frame is
// ~6 passes x 16 B per point), not a tuned one: these sizes are not on BASELINE.json's configs.
#include "generic_fft.h"

namespace {

struct GenArgs {
  const float* x;
  const float* window;
  const float2* tw;      // (N): (cos, -sin)(2 pi k / N)
  float2* out;           // (rows, n_out, M + 1)
  int64_t T, T2, rows, n_out;
  int frame_lo, hop, pad, pad_mode;
  int M, npass;
  int radix[16];
};

struct GenInvArgs {
  const float2* X;       // (rows, n_frames, M + 1)
  const float* window;
  const float2* tw;
  float* frames;         // (rows, n_frames, N)
  int64_t rows, n_frames;
  int M, npass;
  int radix[16];
};

using at::gfft::cmulf;
using at::gfft::dft_r;

// A pointer every lane of the wave agrees on, kept in scalar registers: indexing it with an UNSIGNED 32-bit element
// offset gives the "scalar base + vector offset" addressing mode (one shift per access instead of a 64-bit multiply-add
// chain: the speech-window tiles are VALU-issue bound, logs/s16/pmc.txt)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ src, float2* __restrict__ dst,
                                              const float2* __restrict__ tw, int M, int NS) {
  const int nb = M / R;
  const int tstep = M / (NS * R);          // w_{NS R}^k = w_M^{k tstep}
  for (int j = threadIdx.x; j < nb; j += blockDim.x) {
    const int k = j % NS;
    float2 v[at::gfft::MAX_RADIX];
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = src[j + nb * q];
#pragma unroll
    for (int q = 1; q < R; ++q) v[q] = cmulf(v[q], tw[2 * (k * q * tstep)]);   // table is per N = 2 M
    dft_r<R>(v);
    const int o0 = (j / NS) * NS * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) dst[o0 + NS * q] = v[q];
  }
}

// all passes; returns the buffer holding the result
__device__ __forceinline__ float2* run_passes(float2* a, float2* b, const float2* tw, int M, int npass, const int* radix) {
  int NS = 1;
  for (int p = 0; p < npass; ++p) {
    const int R = radix[p];
    switch (R) {
      case 16: stockham_pass<16>(a, b, tw, M, NS); break;
      case 8: stockham_pass<8>(a, b, tw, M, NS); break;
      case 25: stockham_pass<25>(a, b, tw, M, NS); break;
      case 9: stockham_pass<9>(a, b, tw, M, NS); break;
      case 4: stockham_pass<4>(a, b, tw, M, NS); break;
      case 2: stockham_pass<2>(a, b, tw, M, NS); break;
      case 3: stockham_pass<3>(a, b, tw, M, NS); break;
      case 5: stockham_pass<5>(a, b, tw, M, NS); break;
      default: stockham_pass<7>(a, b, tw, M, NS); break;
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
    NS *= R;
  }
  return a;
}

__global__ __launch_bounds__(256) void stft_generic_kernel(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  const int M = A.M, N = 2 * M;
  float2* bufA = gbuf;
  float2* bufB = gbuf + M;
  const int64_t total = A.rows * A.n_out;
  for (int64_t g = blockIdx.x; g < total; g += gridDim.x) {
    const int64_t row = g / A.n_out;
    const int64_t fo = g - row * A.n_out;
    const float* __restrict__ xr = A.x + row * A.T;
    const int64_t s0 = (fo + A.frame_lo) * (int64_t)A.hop - M;      // first sample, outer-padded coordinates
    const bool interior = A.pad == 0 && s0 >= 0 && s0 + N <= A.T;
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
      float a, b;
      if (interior) { a = xr[s0 + 2 * n]; b = xr[s0 + 2 * n + 1]; }
      else {
        a = at::fetch_padded(xr, s0 + 2 * n, A.T, A.T2, A.pad, A.pad_mode);
        b = at::fetch_padded(xr, s0 + 2 * n + 1, A.T, A.T2, A.pad, A.pad_mode);
      }
      bufA[n] = make_float2(a * A.window[2 * n], b * A.window[2 * n + 1]);
    }
    __syncthreads();
    const float2* Z = run_passes(bufA, bufB, A.tw, M, A.npass, A.radix);
    float2* __restrict__ orow = A.out + g * (M + 1);
    for (int k = threadIdx.x; k <= M / 2; k += blockDim.x) {
      if (k == 0) {
        const float2 z = Z[0];
        orow[0] = make_float2(z.x + z.y, 0.f);
        orow[M] = make_float2(z.x - z.y, 0.f);
      } else {
        const float2 zk = Z[k], zm = Z[M - k];
        const float2 w = A.tw[k];                       // (cos, -sin)(2 pi k / N)
        const float c = w.x, s = -w.y;
        const float sr = zk.x + zm.x, si = zk.y - zm.y;
        const float dr = zk.x - zm.x, di = zk.y + zm.y;
        const float pp = fmaf(s, dr, -c * di);
        const float qq = fmaf(s, di, c * dr);
        orow[k] = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));
        orow[M - k] = make_float2(0.5f * (sr + pp), 0.5f * (-si - qq));
      }
    }
    __syncthreads();
  }
}

// =============================================================================================
// Tiled generic forward kernel (round 3): M = n_fft / 2 <= 4096.
//  * a workgroup transforms FB frames at once (2 for M <= 2048): the batched in-place passes of
//    generic_fft.h keep all 256 threads busy on the radix-16 / 25 passes (one frame of 2048 points
//    has only 128 radix-16 butterflies), one LDS buffer instead of two, composite radices: three
//    passes for 2048 / 4096 points instead of six (PLAN 1 / 2: the pass list is compile-time for the
//    reference's 96 / 192 kHz windows, no run-time dispatch, no plan in SGPRs);
//  * persistent workgroups walk RUNS of consecutive tiles of one row inside their XCD's span of the
//    (row, tile) space, so the 4x overlap between neighbouring frames is served by L1 / that XCD's
//    L2 and HBM sees every sample once (the round-2 kernel dealt frames round-robin over the XCDs:
//    each of them re-read the overlap, input traffic = output traffic);
//  * THE ONLY VECTOR-MEMORY OPERATIONS OF A TILE are the next tile's sample loads (always issued,
//    address clamped, before the passes) and this tile's stores, in that order, and the first tile of a
//    run is peeled: vmcnt retires in order and the compiler counts only what every path issues, so the
//    head of the loop waits for the loads and leaves the stores in flight.  Window, pass twiddles
//    and the mel tables live in LDS; the split twiddle w_N^k is w_M^(k >> 1) times w_N^1 for odd k.
//    (The first version loaded window / twiddles / band weights from L2 behind the stores and guarded
//    the prefetch: every tile began by draining its predecessor's stores -- 24 us per tile, 16 % of HBM.)
//  * every bin X[k], k = 0..M, comes from ONE formula on (Z[k mod M], Z[(M - k) mod M], w_N^k) -- DC and
//    Nyquist included -- in a single ascending sweep: 512-byte store segments, no special cases.  A
//    row with an odd frame count repeats its last-but-one frame in the last tile (same values stored
//    twice) instead of carrying a dead frame;
//  * optional fused mel epilogue (audio_signal.py:1355-1368): |X| goes to LDS (over the transform
//    buffer); the banded filterbank is cut into CHUNKS of 16 bins, every thread forms chunk dot
//    products (weights and magnitudes as ds_read_b128), the partial sums meet in LDS and one thread per
//    (frame, band) adds its band's chunks in order (deterministic).  No second pass over stft_data, no
//    dense matmul.
struct Gen2Args {
  const float* x;
  const float* window;
  const float2* tw;        // (N): (cos, -sin)(2 pi k / N)
  float2* out;             // (rows, n_out, M + 1)
  float* mel;              // (rows, n_out, n_mels) or null
  const int* chunk;        // mel: (n_chunks) first bin of every 16-bin chunk
  const int* band;         // mel: (n_mels, 2) {first chunk, chunk count}
  const float* cw;         // mel: (n_chunks, 16) weights, zero padded
  int64_t T, T2, rows, n_out;
  int64_t tiles_per_row, total_tiles;
  int frame_lo, hop, pad, pad_mode;
  int M, FB, n_mels, n_chunks, run, n_xcd, vec2;
  at::gfft::PassList pl;
};

constexpr int G2_LOADS = at::gfft::TILE_POINTS / 256;     // float2 samples of a tile per thread
constexpr int G2_KB = at::gfft::TILE_POINTS / 256 + 1;    // bins of a tile per thread (FB (M + 1) <= 4098)

#ifndef AT_GENERIC_SMALL_WGS
#define AT_GENERIC_SMALL_WGS 3     // workgroups per CU (168 registers: 4-6 spilled, still 8-12 % faster than two at 174 -- session r04 s09); the register allocation of the speech-window plans must allow
#endif
// PLAN 0: run-time pass list, FB frames per tile;  1: M = 2048 as 16 16 8 (two frames);  2: M = 4096 as 16 16 16;
// round 4, the speech windows with a full tile and a compile-time pass list (no plan in SGPRs, index arithmetic folded):
// 3: n_fft 400 (M = 200 = 25 8, 20 frames);  4: n_fft 1200 (M = 600 = 25 3 8, 6 frames);  5: n_fft 1920 (M = 960 = 5 3 16 4, 4 frames)
template <int PLAN>
__global__ __launch_bounds__(256, (PLAN >= 3 ? AT_GENERIC_SMALL_WGS : 2)) void stft_generic_tiled_kernel(const Gen2Args A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  constexpr bool POW2 = PLAN == 1 || PLAN == 2;        // window in registers, at most two frames per tile
  const int M = PLAN == 1 ? 2048 : PLAN == 2 ? 4096 : PLAN == 3 ? 200 : PLAN == 4 ? 600 : PLAN == 5 ? 960 : A.M;
  const int FB = PLAN == 1 ? 2 : PLAN == 2 ? 1 : PLAN == 3 ? 20 : PLAN == 4 ? 6 : PLAN == 5 ? 4 : A.FB;
  const int N = 2 * M;
  float2* buf = gbuf;                             // [FB][M]
  float2* tw = gbuf + FB * M;                     // [M]: w_M^t
  float2* win = tw + M;                           // [M]: window pairs (PLAN 0; the fixed plans keep theirs in registers)
  float2* stw = !POW2 ? win + M : tw + M;         // [M / 2 + 1]: split twiddles w_N^j
  float* melw = reinterpret_cast<float*>(stw + M / 2 + 2);     // [n_chunks][16]
  int* mtab = reinterpret_cast<int*>(melw + 16 * A.n_chunks);  // [n_chunks] first bins, then [n_mels][2]
  float* part = reinterpret_cast<float*>(mtab + A.n_chunks + 2 * A.n_mels);   // [FB][n_chunks] partial sums
  float* mag = reinterpret_cast<float*>(gbuf);    // [FB][M + 1] (+ slack), over the transform buffer once Z is consumed
  const bool MEL = A.mel != nullptr;
  at::gfft::build_pass_twiddles<256>(tw, A.tw, 2, M, A.pl);        // per-pass blocks of w_M (the table is per N = 2 M)
  // window pairs of this thread's points: n = (tid + 256 i) mod M is the same for every tile
  constexpr int NWR = PLAN == 1 ? 8 : (PLAN == 2 ? 16 : 1);
  float2 wreg[NWR];
  if constexpr (!POW2) {
    for (int i = threadIdx.x; i < M; i += 256) win[i] = reinterpret_cast<const float2*>(A.window)[i];
  } else {
#pragma unroll
    for (int i = 0; i < NWR; ++i) wreg[i] = reinterpret_cast<const float2*>(A.window)[threadIdx.x + 256 * i];
  }
  for (int i = threadIdx.x; i <= M / 2; i += 256) stw[i] = A.tw[i];
  if (MEL) {
    for (int i = threadIdx.x; i < 16 * A.n_chunks; i += 256) melw[i] = A.cw[i];
    for (int i = threadIdx.x; i < A.n_chunks; i += 256) mtab[i] = A.chunk[i];
    for (int i = threadIdx.x; i < 2 * A.n_mels; i += 256) mtab[A.n_chunks + i] = A.band[i];
  }
  __syncthreads();
  using Lay = std::conditional_t<!POW2, at::gfft::RowLayoutN, at::gfft::RowLayout>;
  const Lay lay{M, (M & 15) == 0 ? 1 : 0};
  const int npts = FB * M, nbins = FB * (M + 1);
  const int Ti = (int)A.T, n_out = (int)A.n_out, tpr = (int)A.tiles_per_row, hop = A.hop;
  // frame of a tile index: the fixed plans hold at most two frames; the run-time plan holds FB = up to 64 frames of a
  // short window (n_fft 400: 20 frames per tile -- with two, 16 of the 256 threads had a radix-25 butterfly), so the
  // split is a division by a run-time length: exact through a float reciprocal for indices < 2^13 (the quotient's
  // fractional part stays >= 0.5 / len away from an integer, the product's error is ~1e-3 of that)
  const float inv_M = 1.0f / (float)M, inv_M1 = 1.0f / (float)(M + 1);
  auto frame_of = [&](int e, int len, float inv_len) __attribute__((always_inline)) -> int {
    if constexpr (POW2) return e >= len ? 1 : 0;
    else return (int)(((float)e + 0.5f) * inv_len);
  };

  const int n_x = (int)gridDim.x < A.n_xcd ? (int)gridDim.x : A.n_xcd;
  const int xcd = blockIdx.x % n_x, lblk = blockIdx.x / n_x;
  const int nblk_x = ((int)gridDim.x - xcd + n_x - 1) / n_x;
  const int64_t g_lo = A.total_tiles * xcd / n_x, g_hi = A.total_tiles * (xcd + 1) / n_x;

  // tile g = (row, t): frames f0 .. f0 + FB - 1 with f0 = min(t FB, n_out - FB); s0 = first sample of frame f0
  auto geom = [&](int64_t g, int& row, int& f0, int& s0) {
    row = (int)(g / tpr);
    f0 = min((int)(g - (int64_t)row * tpr) * FB, n_out - FB);
    s0 = (f0 + A.frame_lo) * hop - M;
  };
  auto interior = [&](int s0) { return A.vec2 && A.pad == 0 && s0 >= 0 && s0 + (FB - 1) * hop + N <= Ti; };
  float2 r[G2_LOADS];
  // the samples of a tile: ALWAYS issued (a tile that is not interior, or no tile at all, reads the row start --
  // T >= N + hop is a launch condition of this kernel -- and the values are ignored)
  auto fetch = [&](int64_t g, bool valid, int tid) __attribute__((always_inline)) {
    int row = 0, f0 = 0, s0 = 0;
    if (valid) geom(g, row, f0, s0);
    const bool ok = valid && interior(s0);
    const float* __restrict__ xs = uniform_ptr(A.x + (int64_t)row * A.T + (ok ? s0 : 0));
#pragma unroll
    for (int i = 0; i < G2_LOADS; ++i) {
      if (256 * i < npts) {                               // uniform; lanes past the end repeat the last point
        const int e = min(tid + 256 * i, npts - 1);
        const int fi = frame_of(e, M, inv_M), n = e - fi * M;
        r[i] = *reinterpret_cast<const float2*>(xs + (unsigned)(fi * hop + 2 * n));
      }
    }
  };

  auto tile = [&](int64_t g, int64_t g_end) __attribute__((always_inline)) {
    // opaque copy of the thread index per tile: nothing indexed by it is hoisted out of the persistent loops
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    int row, f0, s0;
    geom(g, row, f0, s0);
    const float* __restrict__ xr = A.x + (int64_t)row * A.T;
    // ---- windowed samples -> LDS
    if (interior(s0)) {
#pragma unroll
      for (int i = 0; i < G2_LOADS; ++i) {
        if (256 * i < npts) {
          const int e = min(tid + 256 * i, npts - 1);
          const int fi = frame_of(e, M, inv_M), n = e - fi * M;
          const float2 w = !POW2 ? win[n] : wreg[i & (NWR - 1)];
          buf[lay.addr(fi, n)] = make_float2(r[i].x * w.x, r[i].y * w.y);
        }
      }
    } else {
#pragma unroll 1
      for (int e = tid; e < npts; e += 256) {
        const int fi = frame_of(e, M, inv_M), n = e - fi * M;
        const int64_t sidx = (int64_t)s0 + fi * hop + 2 * n;
        const float2 w = !POW2 ? win[n] : reinterpret_cast<const float2*>(A.window)[n];
        buf[lay.addr(fi, n)] = make_float2(at::fetch_padded(xr, sidx, A.T, A.T2, A.pad, A.pad_mode) * w.x,
                             at::fetch_padded(xr, sidx + 1, A.T, A.T2, A.pad, A.pad_mode) * w.y);
      }
    }
    __syncthreads();
    fetch(g + 1, g + 1 < g_end, tid);                     // the next tile's samples: in flight during the passes
    if constexpr (PLAN == 1) {
      at::gfft::pass_inplace<16, 1, 256>(buf, tw, 2048, 1, 256, lay);
      at::gfft::pass_inplace<16, 1, 256>(buf, tw, 2048, 16, 256, lay);            // block 0: 16 x 15 entries
      at::gfft::pass_inplace<8, 2, 256>(buf, tw + 240, 2048, 256, 512, lay);      // block 1: 256 x 7
    } else if constexpr (PLAN == 2) {
      at::gfft::pass_inplace<16, 1, 256>(buf, tw, 4096, 1, 256, lay);
      at::gfft::pass_inplace<16, 1, 256>(buf, tw, 4096, 16, 256, lay);
      at::gfft::pass_inplace<16, 1, 256>(buf, tw + 240, 4096, 256, 256, lay);     // block 1: 256 x 15
    } else if constexpr (PLAN == 3) {                     // 20 frames of 200 points: 160 radix-25, 500 radix-8 butterflies
      at::gfft::pass_inplace<25, 1, 256>(buf, tw, 200, 1, 160, lay);
      at::gfft::pass_inplace<8, 2, 256>(buf, tw, 200, 25, 500, lay);              // block 0: 25 x 7
    } else if constexpr (PLAN == 4) {                     // 6 frames of 600 points
      at::gfft::pass_inplace<25, 1, 256>(buf, tw, 600, 1, 144, lay);
      at::gfft::pass_inplace<3, 5, 256>(buf, tw, 600, 25, 1200, lay);             // block 0: 25 x 2
      at::gfft::pass_inplace<8, 2, 256>(buf, tw + 50, 600, 75, 450, lay);         // block 1: 75 x 7
    } else if constexpr (PLAN == 5) {                     // 4 frames of 960 points
      at::gfft::pass_inplace<5, 3, 256>(buf, tw, 960, 1, 768, lay);
      at::gfft::pass_inplace<3, 5, 256>(buf, tw, 960, 5, 1280, lay);              // block 0: 5 x 2
      at::gfft::pass_inplace<16, 1, 256>(buf, tw + 10, 960, 15, 240, lay);        // block 1: 15 x 15
      at::gfft::pass_inplace<4, 4, 256>(buf, tw + 235, 960, 240, 960, lay);       // block 2: 240 x 3
    } else {
      at::gfft::run_passes<256>(buf, tw, M, A.pl, FB, lay);
    }
    // ---- split step; |X| kept for the mel stage
    float mg[G2_KB + 1];                                  // bins (PLAN 1 / 2) or 9 pairs of bins per thread
    float2* __restrict__ orow = uniform_ptr(A.out + ((int64_t)row * n_out + f0) * (M + 1));
    if constexpr (!POW2) {
      // round 4: X[k] and X[M - k] from ONE evaluation of the pair (Z[k], Z[M - k]), k = 0 .. M / 2 per frame (half the
      // LDS reads, index splits and twiddle look-ups of the bin-by-bin sweep below; the quarter table needs no symmetry).
      // mg[2 i], mg[2 i + 1] = |X[k]|, |X[M - k]| of the thread's i-th pair.
      const int hp = M / 2 + 1, npair = FB * hp;
      const float inv_hp = 1.0f / (float)hp;
#pragma unroll
      for (int it = 0; it < G2_KB / 2 + 1; ++it) {
        const int e = tid + 256 * it;
        mg[2 * it] = 0.f; mg[2 * it + 1] = 0.f;
        if (256 * it < npair && e < npair) {
          const int fi = frame_of(e, hp, inv_hp), k = e - fi * hp;
          const float2 zk = buf[lay.addr(fi, k)], zm = buf[lay.addr(fi, k == 0 ? 0 : M - k)];
          const float2 w = stw[k];
          const float c = w.x, sn = -w.y;
          const float sr = zk.x + zm.x, si = zk.y - zm.y;
          const float dr = zk.x - zm.x, di = zk.y + zm.y;
          const float pp = fmaf(sn, dr, -c * di);
          const float qq = fmaf(sn, di, c * dr);
          float2 Xa = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));       // X[k]
          float2 Xb = make_float2(0.5f * (sr + pp), -0.5f * (si + qq));      // X[M - k]
          if (k == 0) { Xa.y = 0.f; Xb.y = 0.f; }           // DC and Nyquist: exactly real
          const unsigned o = (unsigned)(fi * (M + 1) + k);
          orow[o] = Xa;
          if (2 * k != M) orow[o + (unsigned)(M - 2 * k)] = Xb;               // (k = M / 2 is its own partner)
          if (MEL) {
            mg[2 * it] = __builtin_amdgcn_sqrtf(fmaf(Xa.x, Xa.x, Xa.y * Xa.y));
            mg[2 * it + 1] = __builtin_amdgcn_sqrtf(fmaf(Xb.x, Xb.x, Xb.y * Xb.y));
          }
        }
      }
    } else {
#pragma unroll
    for (int it = 0; it < G2_KB; ++it) {
      const int e = tid + 256 * it;
      mg[it] = 0.f;
      if ((POW2 && it < G2_KB - 1) || e < nbins) {        // full-size tiles (PLAN 1 / 2): only the last slot can lie past the tile
        const int ec = min(e, nbins - 1);
        const int fi = frame_of(ec, M + 1, inv_M1), k = ec - fi * (M + 1);
        const float2 zk = buf[lay.addr(fi, k == M ? 0 : k)], zm = buf[lay.addr(fi, (k == 0 || k == M) ? 0 : M - k)];
        // w_N^k from the quarter table stw[j] = w_N^j, j <= M / 2, by the symmetry w_N^(M - j) = -conj(w_N^j)
        float2 w = stw[k <= M / 2 ? k : M - k];
        if (k > M / 2) w = make_float2(-w.x, w.y);
        const float c = w.x, sn = -w.y;
        const float sr = zk.x + zm.x, si = zk.y - zm.y;
        const float dr = zk.x - zm.x, di = zk.y + zm.y;
        const float pp = fmaf(sn, dr, -c * di);
        const float qq = fmaf(sn, di, c * dr);
        float2 X = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));
        if (k == 0 || k == M) X.y = 0.f;                  // exactly real
        if (POW2 && it < G2_KB - 1) orow[e] = X;          // frame f0 + 1 follows frame f0: e indexes both rows
        else if (e < nbins) orow[e] = X;
        if (MEL) mg[it] = __builtin_amdgcn_sqrtf(fmaf(X.x, X.x, X.y * X.y));     // (uniform: no quarter-rate sqrt without a mel stage)
      }
    }
    }
    if (MEL) {
      __syncthreads();                                    // every Z has been read: the buffer becomes |X|
      if constexpr (!POW2) {
        const int hp = M / 2 + 1, npair = FB * hp;
        const float inv_hp = 1.0f / (float)hp;
#pragma unroll
        for (int it = 0; it < G2_KB / 2 + 1; ++it) {
          const int e = tid + 256 * it;
          if (256 * it < npair && e < npair) {
            const int fi = frame_of(e, hp, inv_hp), k = e - fi * hp;
            mag[fi * (M + 1) + k] = mg[2 * it];
            if (2 * k != M) mag[fi * (M + 1) + M - k] = mg[2 * it + 1];
          }
        }
      } else {
#pragma unroll
      for (int it = 0; it < G2_KB; ++it) {
        const int e = tid + 256 * it;
        if (e < nbins) mag[e] = mg[it];
      }
      }
      __syncthreads();
      const int nch = A.n_chunks;
      const float inv_nch = 1.0f / (float)nch, inv_nm = 1.0f / (float)A.n_mels;
#pragma unroll 1
      for (int task = tid; task < FB * nch; task += 256) {
        const int fi = frame_of(task, nch, inv_nch), c = task - fi * nch;
        const float* __restrict__ mrow = mag + fi * (M + 1) + mtab[c];
        const float4* __restrict__ wq = reinterpret_cast<const float4*>(melw + 16 * c);
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 w = wq[q];
          acc = fmaf(w.x, mrow[4 * q], acc);
          acc = fmaf(w.y, mrow[4 * q + 1], acc);
          acc = fmaf(w.z, mrow[4 * q + 2], acc);
          acc = fmaf(w.w, mrow[4 * q + 3], acc);
        }
        part[task] = acc;
      }
      __syncthreads();
      float* __restrict__ mel0 = A.mel + ((int64_t)row * n_out + f0) * A.n_mels;
      for (int task = tid; task < FB * A.n_mels; task += 256) {
        const int fi = frame_of(task, A.n_mels, inv_nm), m = task - fi * A.n_mels;
        const int c0 = mtab[nch + 2 * m], cn = mtab[nch + 2 * m + 1];
        float acc = 0.f;
        for (int c = 0; c < cn; ++c) acc += part[fi * nch + c0 + c];
        mel0[task] = acc;                                 // (frame fi, band m) -> fi * n_mels + m = task
      }
    }
    __syncthreads();                                      // the buffer is free for the next tile
  };

  for (int64_t gbase = g_lo + (int64_t)lblk * A.run; gbase < g_hi; gbase += (int64_t)nblk_x * A.run) {
    const int64_t g_end = min(gbase + A.run, g_hi);
    fetch(gbase, true, (int)threadIdx.x);
    tile(gbase, g_end);                                   // peeled: both edges of the loop below carry "loads, then stores"
    for (int64_t g = gbase + 1; g < g_end; ++g) tile(g, g_end);
  }
}

// =============================================================================================
// The tile of stft_generic_tiled_kernel for its two power-of-two plans (M = 2048: two frames, radices 16 16 8;
// M = 4096: one frame, 16 16 16), written out by hand so that EVERY LDS ADDRESS IS A PER-THREAD BASE PLUS AN
// IMMEDIATE.  The generic passes compute  batch N + (p ^ ((p >> 4) & 15))  per access -- five VALU instructions,
// ~150 accesses per thread and tile: a third of the ~2000 instructions that made the first tiled kernel VALU-issue
// bound (profiles/r03_notes.md).  In the XOR-swizzled row layout a(e) = e ^ ((e >> 4) & 15), e = point of the tile:
//   * offsets that are multiples of 256 commute with the swizzle: sample stores, pass 3, both reads of the split
//     step are  a(t) + 256 i  resp.  a(256 - t) + 256 i;
//   * the reads of the radix-16 passes, points j + (M / 16) q: one base for M = 4096, two (even / odd q) for
//     M = 2048, where the stride 128 flips bit 3 of the swizzle key;
//   * their writes, points 16 j + q and 256 (j >> 4) + 16 q + (j & 15): the key is j & 15 resp. q, i.e. one XOR
//     with a per-thread constant per access.
// One pad slot per 256 points on top of it (slot = a(e) + (e >> 8)): every LDS instruction of the tile touches one
// 256-block per lane group, so the pad is uniform per instruction (no new conflicts), but the reads of a pass are no
// longer multiples of 512 bytes apart and the compiler cannot fuse them into ds_read2st64_b64 -- half the rate of two
// ds_read_b64 and banked mod 32 (MI355X_MICROARCH.md, LDS table); the same trick as PAD256 in fft_wave.h.
// Pass 3 (NS = 256) reads and writes the SAME slots of the same thread: no barrier inside it.  The split step
// forms X[k] and X[M - k] from one evaluation of the pair (Z[k], Z[M - k]) (the generic sweep evaluates every
// bin on its own: twice the arithmetic), k = t + 256 i < M / 2 ascending, M - k descending -- both 512-byte
// segments per wave --, and k = M / 2 by every thread (same value to the same address: the store count of a tile
// stays static).  Everything else -- runs, XCD spans, clamped prefetch, chunked mel epilogue -- is the generic tile.
#define POW2_MAG_STRIDE(M) ((M) + 20)
// float2 slots of the pass-twiddle region.  The mel variant at M = 2048 keeps only w^k, w^2k, w^4k of the third pass
// (the other four powers are one product each): 8 KB less LDS, which is what lets two workgroups share a CU.
#define POW2_TW_SLOTS(PLAN, MELT) (((PLAN) == 1 && (MELT)) ? 240 + 3 * 257 + 13 : ((PLAN) == 1 ? 2048 : 4096))
constexpr int POW2_MEL_SLOTS = 3;     // chunk dot products per thread and tile: FB n_chunks <= 768
// piece sums per band: a band's chunks are consecutive tasks, i.e. lanes of consecutive waves -- at most
// (63 + (M + 1 + 18) / 16) / 64 + 1 of them (a bank with a handful of bands has bands hundreds of bins wide)
#define POW2_MEL_PIECES(PLAN) ((PLAN) == 1 ? 4 : 8)

#ifndef AT_TILED_NT
#define AT_TILED_NT 1            // non-temporal spectrum stores in the hand-addressed forward tile: the write-allocated output
                                 // lines no longer push the overlapping samples of neighbouring tiles out of L2 (n_fft 4096
                                 // 2.59 -> 2.49 ms, 8192 2.71 -> 2.65, two interleaved rounds, s33); 0 = plain stores (A/B)
#endif
template <int PLAN, bool MELT>
__global__ __launch_bounds__(256, 2) void stft_tiled_pow2_kernel(const Gen2Args A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  constexpr int M = PLAN == 1 ? 2048 : 4096, FB = PLAN == 1 ? 2 : 1, N = 2 * M;
  constexpr int R3 = PLAN == 1 ? 8 : 16;          // last radix
  constexpr int NB3 = PLAN == 1 ? 2 : 1;          // pass-3 butterflies per thread (one per frame)
  constexpr int NPI = M / 512;                    // pair iterations per frame: k = t + 256 i < M / 2
  constexpr int NWR = M / 256;                    // distinct window pairs per thread
  constexpr int FS = M + M / 256;                 // slots of a frame incl. its pads
  constexpr int MS = POW2_MAG_STRIDE(M);          // floats of a |X| row: M + 1 bins, zero slack for the last chunk, 16-byte rows
  constexpr bool TW3S = PLAN == 1 && MELT;        // third-pass twiddles from three stored powers
  constexpr bool MEL = MELT;
  constexpr int NSL = POW2_MEL_SLOTS, NPC = POW2_MEL_PIECES(PLAN);
  using at::gfft::dft_r;
  float2* buf = gbuf;                             // [4096 + 16], swizzled + padded
  float2* tw = gbuf + 4096 + 16;                  // pass blocks: [0, 240) pass 2, then rows of 257 for pass 3
  float* magbuf = reinterpret_cast<float*>(tw + POW2_TW_SLOTS(PLAN, MELT));   // [FB][MS] |X| of the PREVIOUS tile (mel only)
  float* melw = magbuf + (MEL ? FB * MS : 0);     // [n_chunks][16] chunk weights
  float* part = melw + 16 * A.n_chunks;           // [FB][n_mels][NPC]: a band's chunk sums, one slot per wave piece
  // pass 2 (NS = 16): tw[(q - 1) 16 + k] = w_256^(k q);  pass 3 (NS = 256): row q - 1 of 257 = w_(256 R3)^(k q) = w_M^(k q)
  // for both plans (the table is per N = 2 M: w_M^x = A.tw[2 x]); rows of 257 for the same reason as the pads.
  // TW3S: rows 0, 1, 2 = w^k, w^2k, w^4k.
  for (int idx = threadIdx.x; idx < 240; idx += 256) {
    const int q1 = idx >> 4, k = idx & 15;
    tw[idx] = A.tw[2 * (k * (q1 + 1) * (M / 256))];
  }
  for (int idx = threadIdx.x; idx < 256 * (TW3S ? 3 : R3 - 1); idx += 256) {
    const int q1 = idx >> 8, k = idx & 255;
    tw[240 + 257 * q1 + k] = A.tw[2 * (k * (TW3S ? (1 << q1) : q1 + 1))];
  }
  // window pairs and split twiddles w_N^k of this thread's points / bins: the same for every tile, in registers
  float2 wreg[NWR], swr[NPI];
#pragma unroll
  for (int i = 0; i < NWR; ++i) wreg[i] = reinterpret_cast<const float2*>(A.window)[threadIdx.x + 256 * i];
#pragma unroll
  for (int i = 0; i < NPI; ++i) swr[i] = A.tw[threadIdx.x + 256 * i];
  const float2 wmid = A.tw[M / 2];
  // ---- mel: this thread's chunk tasks T = t + 256 s (task = frame * n_chunks + chunk), tile-invariant, in registers:
  //      where the 16 magnitudes and weights live, which of the lanes T + 2^i continue the same band inside this
  //      wave (bits 0..5), whether the lane heads its band's piece in this wave (bit 6) and where the piece sum goes;
  //      bit 7: the slot holds a task at all
  int m_off[MEL ? NSL : 1], w_off[MEL ? NSL : 1], m_fl[MEL ? NSL : 1], m_dst[MEL ? NSL : 1];
  if constexpr (MEL) {
    const int nch = A.n_chunks, ntask = FB * nch;
    int* cb = reinterpret_cast<int*>(buf);        // scratch: band of every chunk
    for (int m = threadIdx.x; m < A.n_mels; m += 256) {
      const int c0 = A.band[2 * m], cn = A.band[2 * m + 1];
      for (int c = 0; c < cn; ++c) cb[c0 + c] = m;
    }
    for (int i = threadIdx.x; i < 16 * nch; i += 256) melw[i] = A.cw[i];
    for (int i = threadIdx.x; i < FB * MS; i += 256) magbuf[i] = 0.f;         // the slack behind bin M stays zero
    for (int i = threadIdx.x; i < NPC * FB * A.n_mels; i += 256) part[i] = 0.f; // slots no piece writes stay zero
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s2 = 0; s2 < NSL; ++s2) {
      const int T = (int)threadIdx.x + 256 * s2;
      const bool valid = T < ntask;
      const int Tc = valid ? T : 0;
      const int fi = Tc >= nch ? 1 : 0, c = Tc - fi * nch;
      const int band = cb[c];
      auto same = [&](int T2) {                   // task T2 is a chunk of the same band of the same frame
        if (T2 < 0 || T2 >= ntask) return false;
        const int f2 = T2 >= nch ? 1 : 0;
        return f2 == fi && cb[T2 - f2 * nch] == band;
      };
      int fl = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (valid && lane + (1 << i) < 64 && same(T + (1 << i))) fl |= 1 << i;
      const bool cont = valid && same(T - 1);     // the band began in an earlier task
      if (valid && (lane == 0 || !cont)) fl |= 64;
      m_off[s2] = fi * MS + A.chunk[c];
      w_off[s2] = 16 * c;
      m_fl[s2] = valid ? (fl | 128) : 0;
      // piece = waves between the band's first task and this one
      const int piece = (Tc >> 6) - ((fi * nch + A.band[2 * band]) >> 6);
      m_dst[s2] = NPC * (fi * A.n_mels + band) + (piece < NPC ? piece : NPC - 1);
    }
  }
  __syncthreads();
  const at::gfft::RowLayout lay{M, 1};            // the same layout as a function (edge tiles)
  const int Ti = (int)A.T, n_out = (int)A.n_out, tpr = (int)A.tiles_per_row, hop = A.hop;
  const int n_x = (int)gridDim.x < A.n_xcd ? (int)gridDim.x : A.n_xcd;
  const int xcd = blockIdx.x % n_x, lblk = blockIdx.x / n_x;
  const int nblk_x = ((int)gridDim.x - xcd + n_x - 1) / n_x;
  const int64_t g_lo = A.total_tiles * xcd / n_x, g_hi = A.total_tiles * (xcd + 1) / n_x;

  auto geom = [&](int64_t g, int& row, int& f0, int& s0) {
    row = (int)(g / tpr);
    f0 = min((int)(g - (int64_t)row * tpr) * FB, n_out - FB);
    s0 = (f0 + A.frame_lo) * hop - M;
  };
  auto interior = [&](int s0) { return A.vec2 && A.pad == 0 && s0 >= 0 && s0 + (FB - 1) * hop + N <= Ti; };
  float2 r[16];
  auto fetch = [&](int64_t g, bool valid, int tid) __attribute__((always_inline)) {
    int row = 0, f0 = 0, s0 = 0;
    if (valid) geom(g, row, f0, s0);
    const bool ok = valid && interior(s0);
    const float* __restrict__ xs = A.x + (int64_t)row * A.T + (ok ? s0 : 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i;
      const int fi = PLAN == 1 ? (i >> 3) : 0, n = e - fi * M;
      r[i] = *reinterpret_cast<const float2*>(xs + fi * hop + 2 * n);
    }
  };

  // ---- mel epilogue of a tile, in two halves that ride on the NEXT tile's barriers (the |X| rows have their own LDS
  //      region).  The first version ran behind the split step with three barriers of its own, chunk sums by a loop
  //      and band sums by one thread walking up to 15 chunk sums: +1.05 ms on 2.80 at n_fft 4096, all of it latency.
  //  first half: every thread's <= 3 chunk dot products (16 bins x 16 weights, all reads issued together), then the
  //  chunks of a band are summed across the lanes (they are consecutive tasks; doubling steps through the LDS
  //  crossbar, fixed order) and the head lane of every piece (the part of a band inside one wave) stores its sum.
  auto chunk_dots = [&]() __attribute__((always_inline)) {
    float acc[NSL];
    float4 wv[NSL][4], mv[NSL][4];
#pragma unroll
    for (int s2 = 0; s2 < NSL; ++s2) {
      const float4* __restrict__ mq = reinterpret_cast<const float4*>(magbuf + m_off[s2]);
      const float4* __restrict__ wq = reinterpret_cast<const float4*>(melw + w_off[s2]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { wv[s2][q] = wq[q]; mv[s2][q] = mq[q]; }
    }
#pragma unroll
    for (int s2 = 0; s2 < NSL; ++s2) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a0 = fmaf(wv[s2][q].x, mv[s2][q].x, a0);
        a1 = fmaf(wv[s2][q].y, mv[s2][q].y, a1);
        a0 = fmaf(wv[s2][q].z, mv[s2][q].z, a0);
        a1 = fmaf(wv[s2][q].w, mv[s2][q].w, a1);
      }
      acc[s2] = (m_fl[s2] & 128) ? a0 + a1 : 0.f;                // bit 7: the slot holds a task
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int s2 = 0; s2 < NSL; ++s2) {
        const float sh = __shfl_down(acc[s2], 1 << i, 64);
        acc[s2] += ((m_fl[s2] >> i) & 1) ? sh : 0.f;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < NSL; ++s2)
      if (m_fl[s2] & 64) part[m_dst[s2]] = acc[s2];
  };
  // second half: mel[band] = sum of its pieces in order, stored by one thread per (frame, band) -- ALWAYS (threads past
  // the end repeat the last band): a static store count
  auto band_total = [&](int task) __attribute__((always_inline)) -> float {
    const float4* __restrict__ pp = reinterpret_cast<const float4*>(part + NPC * task);
    float4 p = pp[0];
    float tot = (p.x + p.y) + (p.z + p.w);
    if constexpr (NPC == 8) { p = pp[1]; tot += (p.x + p.y) + (p.z + p.w); }
    return tot;
  };
  auto band_sums = [&](int t, int row, int f0) __attribute__((always_inline)) {
    float* __restrict__ mel0 = A.mel + ((int64_t)row * n_out + f0) * A.n_mels;
    const int ntask = FB * A.n_mels;
    if (ntask <= 256) {
      const int task = min(t, ntask - 1);
      mel0[task] = band_total(task);
    } else {
      for (int task = t; task < ntask; task += 256) mel0[task] = band_total(task);
    }
  };

  int prow = 0, pf0 = 0;                          // the tile whose |X| sits in magbuf
  auto tile = [&](auto pending_c, int64_t g, int64_t g_end) __attribute__((always_inline)) {
    constexpr bool PENDING = decltype(pending_c)::value;      // a previous tile of this run waits for its mel epilogue
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));                   // per tile: nothing indexed by it leaves the persistent loops
    const int st = t ^ ((t >> 4) & 15);           // a(t)
    int row, f0, s0;
    geom(g, row, f0, s0);
    const float* __restrict__ xr = A.x + (int64_t)row * A.T;
    // ---- windowed samples -> LDS
    if (interior(s0)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float2 w = wreg[i & (NWR - 1)];
        buf[st + 257 * i] = make_float2(r[i].x * w.x, r[i].y * w.y);
      }
    } else {
#pragma unroll 1
      for (int e = t; e < 4096; e += 256) {
        const int fi = e >= M ? 1 : 0, n = e - fi * M;
        const int64_t sidx = (int64_t)s0 + fi * hop + 2 * n;
        const float2 w = reinterpret_cast<const float2*>(A.window)[n];
        buf[lay.addr(fi, n) + (e >> 8)] = make_float2(at::fetch_padded(xr, sidx, A.T, A.T2, A.pad, A.pad_mode) * w.x,
                                                      at::fetch_padded(xr, sidx + 1, A.T, A.T2, A.pad, A.pad_mode) * w.y);
      }
    }
    __syncthreads();                              // (also: every |X| of the previous tile is in magbuf)
    fetch(g + 1, g + 1 < g_end, t);               // the next tile's samples: in flight during the passes
    if constexpr (PENDING && MEL) chunk_dots();

    // ---- passes 1 and 2 (radix 16): butterfly j of frame fj
    const int fj = PLAN == 1 ? (t >> 7) : 0, j = PLAN == 1 ? (t & 127) : t;
    const int jl = j & 15;
    const int rb0 = fj * FS + (j ^ ((j >> 4) & 15));                    // reads, even q (M = 4096: every q)
    const int rb1 = PLAN == 1 ? fj * FS + (j ^ ((j >> 4) | 8)) : rb0;   // M = 2048: odd q (stride 128 sets key bit 3)
    const int wb1 = fj * FS + 16 * j + (j >> 4);                        // pass-1 writes: + (q ^ jl); block 16 j >> 8
    const int wb2 = fj * FS + 257 * (j >> 4);                           // pass-2 writes: + 16 q + (jl ^ q)
    // point stride of the reads incl. pads: 128 q + (q >> 1) resp. 257 q
    auto rq = [](int q) constexpr { return PLAN == 1 ? 128 * q + (q >> 1) : 257 * q; };
    float2 v[at::gfft::MAX_RADIX];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = buf[((q & 1) ? rb1 : rb0) + rq(q)];
    dft_r<16>(v);
    __syncthreads();                              // (also: the piece sums of the previous tile are in part[])
    if constexpr (PENDING && MEL) band_sums(t, prow, pf0);
#pragma unroll
    for (int q = 0; q < 16; ++q) buf[wb1 + (q ^ jl)] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = buf[((q & 1) ? rb1 : rb0) + rq(q)];
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmulf(v[q], tw[(q - 1) * 16 + jl]);
    dft_r<16>(v);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) buf[wb2 + 16 * q + (jl ^ q)] = v[q];
    __syncthreads();
    // ---- pass 3 (NS = 256), in place on the thread's own slots
    {
      float2 w3[R3];
      if constexpr (TW3S) {
        const float2 w1 = tw[240 + t], w2 = tw[240 + 257 + t], w4 = tw[240 + 2 * 257 + t];
        const float2 w3_ = cmulf(w1, w2);
        w3[1] = w1; w3[2] = w2; w3[3] = w3_; w3[4] = w4;
        w3[5] = cmulf(w1, w4); w3[6] = cmulf(w2, w4); w3[7] = cmulf(w3_, w4);
      } else {
#pragma unroll
        for (int q = 1; q < R3; ++q) w3[q] = tw[240 + (q - 1) * 257 + t];
      }
#pragma unroll
      for (int b = 0; b < NB3; ++b) {
#pragma unroll
        for (int q = 0; q < R3; ++q) v[q] = buf[st + FS * b + 257 * q];
#pragma unroll
        for (int q = 1; q < R3; ++q) v[q] = cmulf(v[q], w3[q]);
        dft_r<R3>(v);
#pragma unroll
        for (int q = 0; q < R3; ++q) buf[st + FS * b + 257 * q] = v[q];
      }
    }
    __syncthreads();

    // ---- split step: pairs (k, M - k), k = t + 256 i; then k = M / 2.  |X| goes straight to its row buffer.
    const int tm = (256 - t) & 255;                                     // low byte of M - k
    const int sm = (tm ^ ((tm >> 4) & 15)) + (t == 0 ? 257 : 0);        // a(256 - t): t = 0 sits one 256-block (+ its pad) up
    char* __restrict__ obase = reinterpret_cast<char*>(A.out + ((int64_t)row * n_out + f0) * (M + 1));
    auto split = [&](float2 zk, float2 zm, float2 w, float2& xa, float2& xb) __attribute__((always_inline)) {
      const float c = w.x, sn = -w.y;
      const float sr = zk.x + zm.x, si = zk.y - zm.y;
      const float dr = zk.x - zm.x, di = zk.y + zm.y;
      const float pp = fmaf(sn, dr, -c * di);
      const float qq = fmaf(sn, di, c * dr);
      xa = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));             // X[k]
      xb = make_float2(0.5f * (sr + pp), 0.5f * (-si - qq));            // X[M - k]
    };
#pragma unroll
    for (int fi = 0; fi < FB; ++fi) {
      char* __restrict__ ofr = obase + (size_t)fi * (M + 1) * sizeof(float2);
      float* __restrict__ mf = magbuf + fi * MS;
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        const float2 zk = buf[fi * FS + st + 257 * i];
        // Z[M - k]: slot a(256 - t) + 257 (2 NPI - 1 - i); k = 0 pairs with Z[0] itself
        int am = fi * FS + sm + 257 * (2 * NPI - 1 - i);
        if (i == 0) am = t == 0 ? fi * FS : am;
        const float2 zm = buf[am];
        float2 xa, xb;
        split(zk, zm, swr[i], xa, xb);
        if (i == 0) { if (t == 0) { xa.y = 0.f; xb.y = 0.f; } }         // DC and Nyquist: exactly real
        at::stg2<AT_TILED_NT != 0>(reinterpret_cast<float2*>(ofr + (size_t)(t + 256 * i) * sizeof(float2)), xa);
        at::stg2<AT_TILED_NT != 0>(reinterpret_cast<float2*>(ofr + (size_t)(M - 256 * i - 255) * sizeof(float2) + (size_t)(255 - t) * sizeof(float2)), xb);
        if constexpr (MEL) {
          mf[t + 256 * i] = __builtin_amdgcn_sqrtf(fmaf(xa.x, xa.x, xa.y * xa.y));
          mf[(M - 256 * i - 255) + (255 - t)] = __builtin_amdgcn_sqrtf(fmaf(xb.x, xb.x, xb.y * xb.y));
        }
      }
      {
        const float2 z = buf[fi * FS + M / 2 + M / 512];                // a(M / 2) = M / 2, + its pads
        float2 xa, xb;
        split(z, z, wmid, xa, xb);
        *reinterpret_cast<float2*>(ofr + (size_t)(M / 2) * sizeof(float2)) = xa;      // every thread, one address
        if constexpr (MEL) { if (t == 0) mf[M / 2] = __builtin_amdgcn_sqrtf(fmaf(xa.x, xa.x, xa.y * xa.y)); }
      }
    }
    prow = row; pf0 = f0;
    __syncthreads();                                      // every Z has been read: the buffer is free for the next tile
  };

  for (int64_t gbase = g_lo + (int64_t)lblk * A.run; gbase < g_hi; gbase += (int64_t)nblk_x * A.run) {
    const int64_t g_end = min(gbase + A.run, g_hi);
    fetch(gbase, true, (int)threadIdx.x);
    tile(std::false_type{}, gbase, g_end);                // peeled: both edges of the loop below carry "loads, then stores"
    for (int64_t g = gbase + 1; g < g_end; ++g) tile(std::true_type{}, g, g_end);
    if constexpr (MEL) {                                  // the last tile's epilogue (its |X| are complete: barrier above)
      chunk_dots();
      __syncthreads();
      band_sums((int)threadIdx.x, prow, pf0);
    }
  }
}

// =============================================================================================
// Inverse transform for the same two sizes (n_fft 4096 / 8192, hop = n_fft / 4): the hand-addressed tile run backwards,
// with the overlap-add in an LDS ring instead of a (rows, frames, n_fft) frame buffer in HBM + a gather kernel
// (11.8 ms at 256 x 2ch x 10 s @ 96 kHz: 8.8 + 3.1; the forward transform takes 2.7).
//   * a workgroup walks a run of output segments (hops) of one row, two frames (one for M = 4096) per tile, the next
//     tile's bins prefetched into registers; frames outside [lead, lead + n_x) -- match_stride's virtual edge frames,
//     the lead-in of a run, the tail behind the last frame -- load a zero page;
//   * fold: conj(Z[k]) and conj(Z[M - k]) of the half-length transform from one evaluation of the pair (X[k], X[M - k]),
//     k ascending, M - k descending (512-byte segments), straight into the slots the forward split step reads;
//   * the three passes of the forward tile (a forward transform of the conjugate = the inverse); the third leaves
//     point t + 256 q in register q of thread t, which is exactly the sample pair the thread's window register q belongs to;
//   * window (1 / N folded in) and overlap-add: sample pair n of frame f goes to ring slot (f hop / 2 + n) mod M.  Slot s
//     is only ever touched by thread s mod 256 -- accumulate, emit, clear --, so the ring needs no barrier; segment f is
//     complete once frame f has been added and leaves as hop / 512 pair stores per thread, times 1 / envelope;
//   * a fixed number of loads and stores per tile (pairs outside the row go to a dump slot), first tile of a run peeled.
struct GenInv2Args {
  const float2* X;         // (rows, n_x, M + 1)
  const float* window;
  const float2* tw;        // (N): (cos, -sin)(2 pi k / N)
  const float* inv_env;    // ((n_frames - 1) hop + N): 1 / sum_f w^2
  float* dump;             // INV2_DUMP_FLOATS
  const float2* zeros;     // M + 1 zero bins
  float* out;              // (rows, length)
  int64_t rows, length, total_units;
  int n_x, lead, n_frames, n_seg, run, runs_per_row;
};
constexpr int INV2_DUMP_FLOATS = 8 * 256 * 2;      // one pair slot per (emitted pair of a tile, thread)
#define INV2_TW_SLOTS(PLAN) ((PLAN) == 1 ? 2048 : 240 + 4 * 257 + 12)
constexpr int INV2_RUN = 96;                       // segments per run (a multiple of 2); 4 lead-in frames each

__global__ __launch_bounds__(256) void istft_env_generic_kernel(const float* __restrict__ window, float* __restrict__ inv_env,
                                                                int n_frames, int N, int hop, int64_t total, int64_t tail_zero) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gid; i < tail_zero; i += gsz) inv_env[total + i] = 0.f;      // dump slots + zero page behind the table
  for (int64_t pp = gid; pp < total; pp += gsz) {
    int64_t f_hi = pp / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    int64_t f_lo = (pp - N + hop) / hop;
    if (pp - N + 1 <= 0 || f_lo < 0) f_lo = 0;
    float env = 0.f;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
      const int64_t n = pp - f * hop;
      if (n >= 0 && n < N) { const float w = window[n]; env = fmaf(w, w, env); }
    }
    inv_env[pp] = env > 1e-11f ? 1.0f / env : 0.f;
  }
}

template <int PLAN>
__global__ __launch_bounds__(256, 2) void istft_tiled_pow2_kernel(const GenInv2Args A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  constexpr int M = PLAN == 1 ? 2048 : 4096, FB = PLAN == 1 ? 2 : 1, N = 2 * M, HOP = N / 4;
  constexpr int R3 = PLAN == 1 ? 8 : 16;
  constexpr int NPI = M / 512;                    // pair iterations per frame: k = t + 256 i < M / 2
  constexpr int NWR = M / 256;                    // window pairs (= points of a frame) per thread = R3
  constexpr int NSG = HOP / 512;                  // emitted pairs per thread and segment
  constexpr int FS = M + M / 256;
  constexpr int NLD = FB * (2 * NPI + 1);         // bins per thread and tile
  static_assert(NWR == R3, "the third pass leaves a frame's points in the registers of its window pairs");
  // M = 4096 keeps w^k, w^2k, w^4k, w^8k of the third pass's fifteen twiddle rows (the others are one product each):
  // 23 KB less LDS, which is what lets two workgroups share a CU next to the 32 KB ring
  constexpr bool TW3S = PLAN == 2;
  constexpr int TWSLOTS = INV2_TW_SLOTS(PLAN);
  using at::gfft::dft_r;
  float2* buf = gbuf;                             // [4096 + 16]
  float2* tw = gbuf + 4096 + 16;                  // [240 + 257 rows]
  float2* ring = tw + TWSLOTS;                    // [M] overlap-add window, slot = sample pair mod M
  for (int idx = threadIdx.x; idx < 240; idx += 256) {
    const int q1 = idx >> 4, k = idx & 15;
    tw[idx] = A.tw[2 * (k * (q1 + 1) * (M / 256))];
  }
  for (int idx = threadIdx.x; idx < 256 * (TW3S ? 4 : R3 - 1); idx += 256) {
    const int q1 = idx >> 8, k = idx & 255;
    tw[240 + 257 * q1 + k] = A.tw[2 * (k * (TW3S ? (1 << q1) : q1 + 1))];
  }
  float2 wreg[NWR], swr[NPI];
  const float inv_n = 1.0f / (float)N;
#pragma unroll
  for (int i = 0; i < NWR; ++i) {
    const float2 w = reinterpret_cast<const float2*>(A.window)[threadIdx.x + 256 * i];
    wreg[i] = make_float2(w.x * inv_n, w.y * inv_n);
  }
#pragma unroll
  for (int i = 0; i < NPI; ++i) swr[i] = A.tw[threadIdx.x + 256 * i];
  const float2 wmid = A.tw[M / 2];
  __syncthreads();
  const int len = (int)A.length;

  float2 xr_[NLD];                                // the tile's bins: per frame k ascending x NPI, M - k descending x NPI, M / 2
  auto fetch = [&](int64_t row, int f0, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int fi = 0; fi < FB; ++fi) {
      const int f = f0 + fi, fx = f - A.lead;
      const bool live = f >= 0 && f < A.n_frames && fx >= 0 && fx < A.n_x;
      const float2* __restrict__ Xf = live ? A.X + (row * A.n_x + fx) * (int64_t)(M + 1) : A.zeros;
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        xr_[fi * (2 * NPI + 1) + i] = Xf[t + 256 * i];
        xr_[fi * (2 * NPI + 1) + NPI + i] = Xf[(M - 256 * i - 255) + (255 - t)];      // bin M - (t + 256 i)
      }
      xr_[fi * (2 * NPI + 1) + 2 * NPI] = Xf[M / 2];
    }
  };

  for (int64_t unit = blockIdx.x; unit < A.total_units; unit += gridDim.x) {
    const int64_t row = unit / A.runs_per_row;
    const int h0 = (int)(unit - row * A.runs_per_row) * A.run;       // first segment of the run (a multiple of FB)
    const int h1 = min(h0 + A.run, A.n_seg);
    float* __restrict__ orow = A.out + row * A.length;
    for (int i = threadIdx.x; i < M; i += 256) ring[i] = make_float2(0.f, 0.f);   // (slot i belongs to thread i mod 256)
    const int f_first = h0 - 4;                   // lead-in: the three frames before the run reach into its first segment
    fetch(row, f_first, (int)threadIdx.x);

    auto tile = [&](int f0) __attribute__((always_inline)) {
      int t = (int)threadIdx.x;
      asm volatile("" : "+v"(t));
      const int st = t ^ ((t >> 4) & 15);
      const int tm = (256 - t) & 255;
      const int sm = (tm ^ ((tm >> 4) & 15)) + (t == 0 ? 257 : 0);
      // ---- fold the half spectra into the transform buffer: slot of point k := conj(Z[k])
#pragma unroll
      for (int fi = 0; fi < FB; ++fi) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          float2 xk = xr_[fi * (2 * NPI + 1) + i], xm = xr_[fi * (2 * NPI + 1) + NPI + i];
          if (i == 0) { if (t == 0) { xk.y = 0.f; xm.y = 0.f; } }      // c2r ignores the imaginary part of DC and Nyquist
          const float2 w = swr[i];
          const float c = w.x, s = -w.y;
          const float sr = xk.x + xm.x, si = xk.y - xm.y;
          const float dr = xk.x - xm.x, di = xk.y + xm.y;
          const float P = fmaf(s, dr, c * di), Q = fmaf(c, dr, -s * di);
          buf[fi * FS + st + 257 * i] = make_float2(sr - P, -(si + Q));
          // point M - k; for k = 0 there is no point M (its pair slot would be the next frame's first): skipped
          const int am = fi * FS + sm + 257 * (2 * NPI - 1 - i);
          if (!(i == 0 && t == 0)) buf[am] = make_float2(sr + P, si - Q);
        }
        {
          const float2 x = xr_[fi * (2 * NPI + 1) + 2 * NPI];           // k = M / 2 pairs with itself
          const float c = wmid.x, s = -wmid.y;
          const float sr = 2.f * x.x, di = 2.f * x.y;
          const float P = c * di, Q = -s * di;
          if (t == 0) buf[fi * FS + M / 2 + M / 512] = make_float2(sr - P, -Q);
        }
      }
      __syncthreads();
      // reciprocal envelope of the FB segments this tile finishes, THEN the next tile's bins (vmcnt retires in order:
      // the stores below need the envelope, not the prefetch)
      float2 env[FB * NSG];
#pragma unroll
      for (int fi = 0; fi < FB; ++fi) {
        const int f = f0 + fi;
        const int fe = f < 0 ? 0 : (f >= A.n_seg ? A.n_seg - 1 : f);
        const float2* __restrict__ e2 = reinterpret_cast<const float2*>(A.inv_env) + (int64_t)fe * (HOP / 2);
#pragma unroll
        for (int jj = 0; jj < NSG; ++jj) env[fi * NSG + jj] = e2[t + 256 * jj];
      }
      fetch(row, f0 + FB, t);
      // ---- the forward tile's passes
      const int fj = PLAN == 1 ? (t >> 7) : 0, j = PLAN == 1 ? (t & 127) : t;
      const int jl = j & 15;
      const int rb0 = fj * FS + (j ^ ((j >> 4) & 15));
      const int rb1 = PLAN == 1 ? fj * FS + (j ^ ((j >> 4) | 8)) : rb0;
      const int wb1 = fj * FS + 16 * j + (j >> 4);
      const int wb2 = fj * FS + 257 * (j >> 4);
      auto rq = [](int q) constexpr { return PLAN == 1 ? 128 * q + (q >> 1) : 257 * q; };
      float2 v[at::gfft::MAX_RADIX];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = buf[((q & 1) ? rb1 : rb0) + rq(q)];
      dft_r<16>(v);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) buf[wb1 + (q ^ jl)] = v[q];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = buf[((q & 1) ? rb1 : rb0) + rq(q)];
#pragma unroll
      for (int q = 1; q < 16; ++q) v[q] = cmulf(v[q], tw[(q - 1) * 16 + jl]);
      dft_r<16>(v);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) buf[wb2 + 16 * q + (jl ^ q)] = v[q];
      __syncthreads();
      // ---- third pass per frame, then window, overlap-add and the finished segment
      float2 w3[R3];
      if constexpr (TW3S) {
        const float2 w1 = tw[240 + t], w2 = tw[240 + 257 + t], w4 = tw[240 + 2 * 257 + t], w8 = tw[240 + 3 * 257 + t];
        w3[1] = w1; w3[2] = w2; w3[4] = w4; w3[8] = w8;
        w3[3] = cmulf(w1, w2); w3[5] = cmulf(w1, w4); w3[6] = cmulf(w2, w4); w3[7] = cmulf(w3[3], w4);
#pragma unroll
        for (int q = 9; q < 16; ++q) w3[q] = cmulf(w3[q - 8], w8);
      } else {
#pragma unroll
        for (int q = 1; q < R3; ++q) w3[q] = tw[240 + (q - 1) * 257 + t];
      }
#pragma unroll
      for (int b = 0; b < FB; ++b) {
#pragma unroll
        for (int q = 0; q < R3; ++q) v[q] = buf[st + FS * b + 257 * q];
#pragma unroll
        for (int q = 1; q < R3; ++q) v[q] = cmulf(v[q], w3[q]);
        dft_r<R3>(v);
        const int f = f0 + b;
        const int base = ((f & 3) * (HOP / 2) + t) & (M - 1);            // ring slot of pair n = t; pair t + 256 q: + 256 q
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          const int slot = (base + 256 * q) & (M - 1);
          float2 a = ring[slot];
          a.x = fmaf(v[q].x, wreg[q].x, a.x);
          a.y = fmaf(-v[q].y, wreg[q].y, a.y);
          ring[slot] = a;
        }
        // segment f = samples [f HOP, (f + 1) HOP) of the padded row: pairs t + 256 jj, ring slots base + 256 jj
        const bool emit = f >= h0 && f < h1;
        bool cut = false;
#pragma unroll
        for (int jj = 0; jj < NSG; ++jj) {
          const int slot = (base + 256 * jj) & (M - 1);
          const float2 a = ring[slot];
          ring[slot] = make_float2(0.f, 0.f);
          const float2 e = env[b * NSG + jj];
          const int p = f * HOP + 2 * (t + 256 * jj) - N / 2;            // output sample of the pair (centre padding removed)
          const bool full = emit && p >= 0 && p + 1 < len;
          cut |= emit && !full && p + 1 >= 0 && p < len;
          float* dst = full ? orow + p : A.dump + ((b * NSG + jj) * 256 + t) * 2;
          at::stg2_a4<false>(dst, a.x * e.x, a.y * e.y);
          if (!full) { v[jj].x = a.x * e.x; v[jj].y = a.y * e.y; }       // kept for the element stores below
        }
        if (__any(cut)) {            // a pair cut by the end of an odd-length row (or by its start): element stores
#pragma unroll
          for (int jj = 0; jj < NSG; ++jj) {
            const int p = f * HOP + 2 * (t + 256 * jj) - N / 2;
            if (!emit || (p >= 0 && p + 1 < len)) continue;
            if (p >= 0 && p < len) orow[p] = v[jj].x;
            if (p + 1 >= 0 && p + 1 < len) orow[p + 1] = v[jj].y;
          }
        }
      }
      __syncthreads();               // every read of the transform buffer is done: the next fold may overwrite it
    };

    tile(f_first);                   // peeled: both edges of the loop below carry "loads, then stores"
    for (int f0 = f_first + FB; f0 < h1; f0 += FB) tile(f0);
  }
}

// Inverse frames, many per tile (round 4): the speech windows (n_fft 400 / 1200 / 1920) spent 4.3-6.8 ms per 512 x 10 s rows
// in the one-frame-per-workgroup kernel below (200 of 256 threads fold, 8-24 carry a butterfly, six barriers per frame)
// against 0.7-2.5 ms for the forward tile.  Same cure as the forward: FB consecutive frames (frames are contiguous in the
// spectrum AND in the frame buffer, across rows too, so a tile is one contiguous read and one contiguous write) share the
// in-place batched passes of generic_fft.h.  a[k] = conj(Z[k]) -> forward FFT -> conj: istft_frames_kernel's identity.
struct GenInvTileArgs {
  const float2* X;       // (total, M + 1)
  const float* window;
  const float2* tw;      // (N): (cos, -sin)(2 pi k / N)
  float* frames;         // (total, N)
  int64_t total;         // rows * n_frames
  int M, FB;
  at::gfft::PassList pl;
};

constexpr int GI_LOADS = at::gfft::TILE_POINTS / 256;

template <int PLAN>
__global__ __launch_bounds__(256, 3) void istft_frames_generic_tiled_kernel(const GenInvTileArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  const int M = PLAN == 3 ? 200 : PLAN == 4 ? 600 : PLAN == 5 ? 960 : A.M;
  const int FB = PLAN == 3 ? 20 : PLAN == 4 ? 6 : PLAN == 5 ? 4 : A.FB;
  float2* buf = gbuf;                 // [FB][M]
  float2* tw = gbuf + FB * M;         // [M]: pass blocks of w_M
  float2* win = tw + M;               // [M]: window pairs
  float2* ftw = win + M;              // [M]: fold twiddles w_N^k
  at::gfft::build_pass_twiddles<256>(tw, A.tw, 2, M, A.pl);
  for (int i = threadIdx.x; i < M; i += 256) {
    win[i] = reinterpret_cast<const float2*>(A.window)[i];
    ftw[i] = A.tw[i];
  }
  __syncthreads();
  const at::gfft::RowLayoutN lay{M, (M & 15) == 0 ? 1 : 0};
  const int npts = FB * M;
  const float inv_M = 1.0f / (float)M, inv_n = 1.0f / (float)(2 * M);
  const int64_t n_tiles = (A.total + FB - 1) / FB;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    // the last tile is moved back to hold FB whole frames (its first frames are written twice, same values)
    const int64_t g0 = min(t * FB, A.total - FB);
    const float2* __restrict__ Xt = A.X + g0 * (M + 1);
    float2 xa[GI_LOADS], xm[GI_LOADS];
#pragma unroll
    for (int i = 0; i < GI_LOADS; ++i) {
      if (256 * i < npts) {                                 // uniform; lanes past the end repeat the last point
        const int e = min(tid + 256 * i, npts - 1);
        const int fi = (int)(((float)e + 0.5f) * inv_M), k = e - fi * M;
        const float2* __restrict__ Xf = Xt + fi * (M + 1);
        xa[i] = Xf[k];
        xm[i] = Xf[M - k];
      }
    }
#pragma unroll
    for (int i = 0; i < GI_LOADS; ++i) {
      if (256 * i < npts) {
        const int e = min(tid + 256 * i, npts - 1);
        const int fi = (int)(((float)e + 0.5f) * inv_M), k = e - fi * M;
        float2 a = xa[i], m = xm[i];
        if (k == 0) { a.y = 0.f; m.y = 0.f; }               // c2r ignores the imaginary part of DC and Nyquist
        const float2 w = ftw[k];
        const float c = w.x, s = -w.y;
        const float sr = a.x + m.x, si = a.y - m.y;
        const float dr = a.x - m.x, di = a.y + m.y;
        const float zr = sr - s * dr - c * di;
        const float zi = si + c * dr - s * di;
        buf[lay.addr(fi, k)] = make_float2(zr, -zi);
      }
    }
    __syncthreads();
    if constexpr (PLAN == 3) {
      at::gfft::pass_inplace<25, 1, 256>(buf, tw, 200, 1, 160, lay);
      at::gfft::pass_inplace<8, 2, 256>(buf, tw, 200, 25, 500, lay);
    } else if constexpr (PLAN == 4) {
      at::gfft::pass_inplace<25, 1, 256>(buf, tw, 600, 1, 144, lay);
      at::gfft::pass_inplace<3, 5, 256>(buf, tw, 600, 25, 1200, lay);
      at::gfft::pass_inplace<8, 2, 256>(buf, tw + 50, 600, 75, 450, lay);
    } else if constexpr (PLAN == 5) {
      at::gfft::pass_inplace<5, 3, 256>(buf, tw, 960, 1, 768, lay);
      at::gfft::pass_inplace<3, 5, 256>(buf, tw, 960, 5, 1280, lay);
      at::gfft::pass_inplace<16, 1, 256>(buf, tw + 10, 960, 15, 240, lay);
      at::gfft::pass_inplace<4, 4, 256>(buf, tw + 235, 960, 240, 960, lay);
    } else {
      at::gfft::run_passes<256>(buf, tw, M, A.pl, FB, lay);
    }
    float2* __restrict__ out = reinterpret_cast<float2*>(A.frames) + g0 * M;
#pragma unroll
    for (int i = 0; i < GI_LOADS; ++i) {
      const int e = tid + 256 * i;
      if (e < npts) {
        const int fi = (int)(((float)e + 0.5f) * inv_M), n = e - fi * M;
        const float2 y = buf[lay.addr(fi, n)];
        const float2 w = win[n];
        out[e] = make_float2(y.x * inv_n * w.x, -y.y * inv_n * w.y);
      }
    }
    __syncthreads();
  }
}

// Inverse transform of the run-time sizes in ONE pass (round 4): the tile above, with the overlap-add done in LDS instead
// of through a (rows, frames, n_fft) buffer in HBM and a gather kernel (n_fft 400: 2 x 1.3 GB of frame traffic for 0.33 GB
// of audio).  A workgroup walks a run of consecutive tiles of one row.  The LDS keeps R + FB frame slots in a row: the
// last R = ceil(N / hop) - 1 transformed frames of the previous tile ("history") in front of the FB frames of this one, so
// every output sample of the tile's FB hops -- centre-padded positions [f0 hop, (f0 + FB) hop) -- finds all its <= R + 1
// frames in LDS: sum in ascending frame order (as istft_ola_kernel), window and 1 / N folded into one LDS table, times
// 1 / envelope (table of istft_env_generic_kernel), one coalesced store.  Then the last R slots move to the front.
// Frames outside [0, n_frames) are zero spectra (nothing is loaded); a run that starts inside a row first transforms
// the tile in front of it for its history.  hop and n_fft / 2 even (sample PAIRS stay aligned between frames), FB >= R.
struct GenOlaArgs {
  const float2* X;         // (rows, n_frames, M + 1)
  const float* window;
  const float2* tw;
  const float* inv_env;    // (env_n)
  float* out;              // (rows, length)
  int64_t rows, length, env_n, total_runs;
  int n_frames, M, FB, R, h2 /* hop / 2 */, tiles_per_row, runs_per_row, tiles_per_run, vec2;
  at::gfft::PassList pl;
};

template <int PLAN>
__global__ __launch_bounds__(256, 2) void istft_generic_ola_kernel(const GenOlaArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  const int M = PLAN == 3 ? 200 : PLAN == 4 ? 600 : PLAN == 5 ? 960 : A.M;
  const int FB = PLAN == 3 ? 20 : PLAN == 4 ? 6 : PLAN == 5 ? 4 : A.FB;
  const int R = A.R, h2 = A.h2;
  float2* slots = gbuf;               // [R + FB][M]: history, then this tile's frames
  float2* buf = slots + R * M;
  float2* tw = buf + FB * M;          // [M]: pass blocks of w_M
  float2* win2 = tw + M;              // [M]: (w[2n] / N, -w[2n+1] / N): the conj of the transform's identity folded in
  float2* ftw = win2 + M;             // [M]: fold twiddles w_N^k
  at::gfft::build_pass_twiddles<256>(tw, A.tw, 2, M, A.pl);
  {
    const float inv_n = 1.0f / (float)(2 * M);
    for (int i = threadIdx.x; i < M; i += 256) {
      const float2 w = reinterpret_cast<const float2*>(A.window)[i];
      win2[i] = make_float2(w.x * inv_n, -w.y * inv_n);
      ftw[i] = A.tw[i];
    }
  }
  __syncthreads();
  const at::gfft::RowLayoutN lay{M, (M & 15) == 0 ? 1 : 0};
  const int npts = FB * M, npairs = FB * h2;
  const float inv_M = 1.0f / (float)M, inv_h2 = 1.0f / (float)h2;
  const int half = M / 2;             // output sample p = 2 j - M: pair j - M / 2 of the row (M even)

  auto transform = [&](int64_t row, int f0) __attribute__((always_inline)) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const float2* __restrict__ Xt = A.X + (row * A.n_frames + f0) * (int64_t)(M + 1);
    float2 xa[GI_LOADS], xm[GI_LOADS];
#pragma unroll
    for (int i = 0; i < GI_LOADS; ++i) {
      if (256 * i < npts) {
        const int e = min(tid + 256 * i, npts - 1);
        const int fi = (int)(((float)e + 0.5f) * inv_M), k = e - fi * M;
        xa[i] = make_float2(0.f, 0.f);
        xm[i] = make_float2(0.f, 0.f);
        if (f0 + fi < A.n_frames) {
          const float2* __restrict__ Xf = Xt + fi * (M + 1);
          xa[i] = Xf[k];
          xm[i] = Xf[M - k];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < GI_LOADS; ++i) {
      if (256 * i < npts) {
        const int e = min(tid + 256 * i, npts - 1);
        const int fi = (int)(((float)e + 0.5f) * inv_M), k = e - fi * M;
        float2 a = xa[i], m = xm[i];
        if (k == 0) { a.y = 0.f; m.y = 0.f; }
        const float2 w = ftw[k];
        const float c = w.x, s = -w.y;
        const float sr = a.x + m.x, si = a.y - m.y;
        const float dr = a.x - m.x, di = a.y + m.y;
        const float zr = sr - s * dr - c * di;
        const float zi = si + c * dr - s * di;
        buf[lay.addr(fi, k)] = make_float2(zr, -zi);
      }
    }
    __syncthreads();
    if constexpr (PLAN == 3) {
      at::gfft::pass_inplace<25, 1, 256>(buf, tw, 200, 1, 160, lay);
      at::gfft::pass_inplace<8, 2, 256>(buf, tw, 200, 25, 500, lay);
    } else if constexpr (PLAN == 4) {
      at::gfft::pass_inplace<25, 1, 256>(buf, tw, 600, 1, 144, lay);
      at::gfft::pass_inplace<3, 5, 256>(buf, tw, 600, 25, 1200, lay);
      at::gfft::pass_inplace<8, 2, 256>(buf, tw + 50, 600, 75, 450, lay);
    } else if constexpr (PLAN == 5) {
      at::gfft::pass_inplace<5, 3, 256>(buf, tw, 960, 1, 768, lay);
      at::gfft::pass_inplace<3, 5, 256>(buf, tw, 960, 5, 1280, lay);
      at::gfft::pass_inplace<16, 1, 256>(buf, tw + 10, 960, 15, 240, lay);
      at::gfft::pass_inplace<4, 4, 256>(buf, tw + 235, 960, 240, 960, lay);
    } else {
      at::gfft::run_passes<256>(buf, tw, M, A.pl, FB, lay);
    }
  };
  auto keep_history = [&]() __attribute__((always_inline)) {      // slots [FB, FB + R) -> [0, R); FB >= R: no overlap
    for (int i = threadIdx.x; i < R * M; i += 256) slots[i] = slots[FB * M + i];
    __syncthreads();
  };

  for (int64_t run = blockIdx.x; run < A.total_runs; run += gridDim.x) {
    const int64_t row = run / A.runs_per_row;
    const int t_first = (int)(run - row * A.runs_per_row) * A.tiles_per_run;
    const int t_last = min(t_first + A.tiles_per_run, A.tiles_per_row);
    if (t_first == 0) {
      for (int i = threadIdx.x; i < R * M; i += 256) slots[i] = make_float2(0.f, 0.f);
      __syncthreads();
    } else {
      transform(row, (t_first - 1) * FB);
      keep_history();
    }
    float* __restrict__ orow = A.out + row * A.length;
    for (int t = t_first; t < t_last; ++t) {
      const int f0 = t * FB;
      transform(row, f0);
      // pair jj of the tile: centre-padded samples 2 (f0 h2 + jj), +1; newest frame q = jj / h2, then q - 1 ... q - R
      const int64_t j0 = (int64_t)f0 * h2;
      for (int jj = threadIdx.x; jj < npairs; jj += 256) {
        const int q = (int)(((float)jj + 0.5f) * inv_h2);
        const int rem = jj - q * h2;
        float2 acc = make_float2(0.f, 0.f);
        for (int k = R; k >= 0; --k) {                          // ascending frame order
          const int n2 = rem + k * h2;
          if (n2 < M) {
            const float2 v = slots[lay.addr(R + q - k, n2)];
            const float2 w = win2[n2];
            acc.x = fmaf(v.x, w.x, acc.x);
            acc.y = fmaf(v.y, w.y, acc.y);
          }
        }
        const int64_t j = j0 + jj;
        const int64_t p = 2 * j - M;
        if (A.vec2) {
          if (p >= 0 && p + 1 < A.length) {
            float2 ie = make_float2(0.f, 0.f);
            if (2 * j + 1 < A.env_n) ie = *reinterpret_cast<const float2*>(A.inv_env + 2 * j);
            *reinterpret_cast<float2*>(orow + p) = make_float2(acc.x * ie.x, acc.y * ie.y);
          }
        } else {
          if (p >= 0 && p < A.length) orow[p] = 2 * j < A.env_n ? acc.x * A.inv_env[2 * j] : 0.f;
          if (p + 1 >= 0 && p + 1 < A.length) orow[p + 1] = 2 * j + 1 < A.env_n ? acc.y * A.inv_env[2 * j + 1] : 0.f;
        }
      }
      (void)half;
      __syncthreads();
      keep_history();
    }
  }
}

__global__ __launch_bounds__(256) void istft_frames_generic_kernel(const GenInvArgs A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  const int M = A.M, N = 2 * M;
  float2* bufA = gbuf;
  float2* bufB = gbuf + M;
  const int64_t total = A.rows * A.n_frames;
  const float inv_n = 1.0f / (float)N;
  for (int64_t g = blockIdx.x; g < total; g += gridDim.x) {
    const float2* __restrict__ Xf = A.X + g * (M + 1);
    // fold the half spectrum: a[k] = conj(Z[k]) (istft.hip istft_frames_kernel)
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
      float2 xa = Xf[k], xm = Xf[M - k];
      if (k == 0) { xa.y = 0.f; xm.y = 0.f; }           // c2r ignores the imaginary part of DC and Nyquist
      const float2 w = A.tw[k];
      const float c = w.x, s = -w.y;
      const float sr = xa.x + xm.x, si = xa.y - xm.y;
      const float dr = xa.x - xm.x, di = xa.y + xm.y;
      const float zr = sr - s * dr - c * di;
      const float zi = si + c * dr - s * di;
      bufA[k] = make_float2(zr, -zi);
    }
    __syncthreads();
    const float2* Y = run_passes(bufA, bufB, A.tw, M, A.npass, A.radix);
    float2* __restrict__ out = reinterpret_cast<float2*>(A.frames + g * N);
    const float2* __restrict__ w2 = reinterpret_cast<const float2*>(A.window);
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
      const float2 y = Y[n];
      const float2 w = w2[n];
      out[n] = make_float2(y.x * inv_n * w.x, -y.y * inv_n * w.y);
    }
    __syncthreads();
  }
}

}  // namespace

namespace at {

int generic_fft_plan(int n_fft, int* radix) {
  if (n_fft < 4 || n_fft > 16384 || (n_fft & 1)) return 0;
  int m = n_fft / 2, n = 0;
  // composite radices first: 2048 = 16 * 16 * 8 is three passes (barriers) instead of 4^5 * 2 = six
  for (int r : {16, 8, 4, 2, 25, 5, 9, 3, 7})
    while (m % r == 0 && n < 16) { radix[n++] = r; m /= r; }
  if (m != 1) return 0;
  if (n == 0) { radix[0] = 1; return 0; }     // n_fft == 2: not worth a kernel
  return n;
}

static int grid_for(int64_t frames) {
  const int64_t cap = (int64_t)device_cu_count() * 8;
  return (int)(frames < cap ? frames : cap);
}

int stft_generic(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles, int n_fft, int hop,
                 int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out, float* stft_out,
                 const int* mel_chunk, const int* mel_band, const float* mel_w, int n_chunks, int n_mels, float* mel_out,
                 hipStream_t st) {
  static const int force_old = env_int_once("AT_STFT_GENERIC_OLD", 0);    // A/B: the round-2 one-frame-per-workgroup kernel
  const int M = n_fft / 2;
  // frames per tile: two for the hand-addressed M = 2048 plan, one above it; the run-time plan fills its 4096-point
  // tile (round 4: n_fft 400 -> 20 frames, 1200 -> 6, 1920 -> 4; with two, a radix-25 pass of n_fft 400 had 16 of 256
  // threads busy and the tile's barriers were most of its time), within what the row holds
  int FBv = M <= gfft::TILE_POINTS / 2 ? 2 : 1;
  static const int fb_cap = env_int_once("AT_STFT_GENERIC_FB", 64);       // A/B: 2 = the round-3 tile
  if (M != 2048 && M <= gfft::TILE_POINTS / 2) {
    int fb = gfft::TILE_POINTS / M;
    if (fb > (gfft::TILE_POINTS + 2) / (M + 1)) fb = (gfft::TILE_POINTS + 2) / (M + 1);
    if (fb > fb_cap) fb = fb_cap;
    if (fb > n_frames_out) fb = (int)n_frames_out;
    while (fb > 1 && T < (int64_t)n_fft + (int64_t)(fb - 1) * hop) --fb;
    if (mel_out) while (fb > 2 && ((size_t)fb * M + 2 * (size_t)M + M / 2 + 2) * 8 + 64 + (size_t)n_chunks * (16 + 1 + fb) * 4 + (size_t)2 * n_mels * 4 > 160 * 1024) --fb;
    FBv = fb < 1 ? 1 : fb;
  }
  // the tiled kernel needs FB whole frames per row and an interior position for its clamped prefetch
  const bool tiled_ok = M <= gfft::TILE_POINTS && n_frames_out >= FBv && T >= (int64_t)n_fft + (int64_t)(FBv - 1) * hop &&
                        (int64_t)n_frames_out * hop + n_fft < (1LL << 31);
  if (tiled_ok && !(force_old && mel_out == nullptr)) {
    Gen2Args G;
    if (!gfft::factor(M, &G.pl)) return AT_ERR_UNSUPPORTED;
    G.x = x; G.window = window; G.tw = reinterpret_cast<const float2*>(twiddles); G.out = reinterpret_cast<float2*>(stft_out);
    G.mel = mel_out; G.chunk = mel_chunk; G.band = mel_band; G.cw = mel_w;
    G.n_mels = mel_out ? n_mels : 0; G.n_chunks = mel_out ? n_chunks : 0;
    G.T = T; G.T2 = T + 2 * (int64_t)pad + right_pad; G.rows = rows; G.n_out = n_frames_out;
    G.frame_lo = frame_lo; G.hop = hop; G.pad = pad; G.pad_mode = pad_mode; G.M = M; G.FB = FBv;
    G.tiles_per_row = (n_frames_out + G.FB - 1) / G.FB;
    G.total_tiles = rows * G.tiles_per_row;
    G.vec2 = ((T % 2) == 0 && (hop % 2) == 0 && (M % 2) == 0 && (reinterpret_cast<uintptr_t>(x) % 8) == 0) ? 1 : 0;
    static const int small_plans = env_int_once("AT_STFT_GENERIC_PLANS", 1);     // A/B: 0 = run-time plan for the speech windows
    const int plan = (M == 2048 && G.FB == 2) ? 1 : (M == 4096 ? 2 : 0);
    // compile-time pass lists of the speech windows (full tiles only; their factorisation is what gfft::factor returns)
    const int splan = !small_plans ? 0 : (M == 200 && G.FB == 20) ? 3 : (M == 600 && G.FB == 6) ? 4 : (M == 960 && G.FB == 4) ? 5 : 0;
    // transform buffer (+ |X| slack), pass twiddles, [window: run-time plans only], split twiddles
    static const int old_tile = env_int_once("AT_STFT_TILED_OLD", 0);      // A/B: the generic tile for the fixed plans
    // hand-addressed tile (its mel stage holds <= 3 chunk tasks per thread)
    const bool pow2 = plan != 0 && !old_tile && (!mel_out || (int64_t)G.FB * G.n_chunks <= 256 * POW2_MEL_SLOTS);
    size_t lds = ((size_t)G.FB * M + (plan == 0 ? 2 : 1) * (size_t)M + M / 2 + 2) * sizeof(float2) + 64;
    if (mel_out) lds += (size_t)G.n_chunks * (16 + 1 + G.FB) * 4 + (size_t)2 * n_mels * 4;
    // padded transform buffer, pass twiddles; window and split twiddles in registers; mel: |X| rows, chunk weights, piece sums
    if (pow2) {
      lds = ((size_t)4096 + 16 + (mel_out ? POW2_TW_SLOTS(plan, true) : POW2_TW_SLOTS(plan, false))) * sizeof(float2) + 64;
      if (mel_out) lds += ((size_t)G.FB * POW2_MAG_STRIDE(M) + (size_t)16 * G.n_chunks + (size_t)POW2_MEL_PIECES(plan) * G.FB * n_mels) * 4;
    }
    if (lds > 160 * 1024) return AT_ERR_UNSUPPORTED;
    const void* kfn = pow2 ? (plan == 1 ? (mel_out ? reinterpret_cast<const void*>(stft_tiled_pow2_kernel<1, true>)
                                                   : reinterpret_cast<const void*>(stft_tiled_pow2_kernel<1, false>))
                                        : (mel_out ? reinterpret_cast<const void*>(stft_tiled_pow2_kernel<2, true>)
                                                   : reinterpret_cast<const void*>(stft_tiled_pow2_kernel<2, false>)))
                    : plan == 1 ? reinterpret_cast<const void*>(stft_generic_tiled_kernel<1>)
                    : plan == 2 ? reinterpret_cast<const void*>(stft_generic_tiled_kernel<2>)
                    : splan == 3 ? reinterpret_cast<const void*>(stft_generic_tiled_kernel<3>)
                    : splan == 4 ? reinterpret_cast<const void*>(stft_generic_tiled_kernel<4>)
                    : splan == 5 ? reinterpret_cast<const void*>(stft_generic_tiled_kernel<5>)
                                : reinterpret_cast<const void*>(stft_generic_tiled_kernel<0>);
    int e = allow_big_lds(kfn);
    if (e != AT_OK) return e;
    const int n_cu = device_cu_count();
    int per_cu = (int)((160 * 1024) / lds);
    // the hand-addressed tile needs ~155 registers: three workgroups (12 waves) per CU where the LDS allows it (M = 2048 without mel)
    static const int cap_env = env_int_once("AT_STFT_TILED_WGS", 0);         // A/B
    const int cap = cap_env > 0 ? cap_env : (pow2 ? 3 : (splan ? AT_GENERIC_SMALL_WGS : 2));
    per_cu = per_cu > cap ? cap : (per_cu < 1 ? 1 : per_cu);
    int64_t blocks = (int64_t)n_cu * per_cu;
    if (blocks > G.total_tiles) blocks = G.total_tiles;
    int auto_x = n_cu / 32;
    G.n_xcd = auto_x < 1 ? 1 : (auto_x > 8 ? 8 : auto_x);
    // runs: every workgroup of an XCD span gets the same number of whole runs of <= 32 tiles
    {
      const int64_t n_x = blocks < G.n_xcd ? blocks : G.n_xcd;
      const int64_t wg_x = blocks / n_x > 0 ? blocks / n_x : 1;
      const int64_t span = (G.total_tiles + n_x - 1) / n_x;
      const int64_t per_wg = (span + wg_x - 1) / wg_x;
      const int64_t runs = (per_wg + 31) / 32;
      int64_t run = (per_wg + runs - 1) / (runs > 0 ? runs : 1);
      G.run = (int)(run < 1 ? 1 : run);
    }
    if (pow2 && plan == 1 && mel_out) hipLaunchKernelGGL((stft_tiled_pow2_kernel<1, true>), dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (pow2 && plan == 1) hipLaunchKernelGGL((stft_tiled_pow2_kernel<1, false>), dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (pow2 && mel_out) hipLaunchKernelGGL((stft_tiled_pow2_kernel<2, true>), dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (pow2) hipLaunchKernelGGL((stft_tiled_pow2_kernel<2, false>), dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (plan == 1) hipLaunchKernelGGL(stft_generic_tiled_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (plan == 2) hipLaunchKernelGGL(stft_generic_tiled_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (splan == 3) hipLaunchKernelGGL(stft_generic_tiled_kernel<3>, dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (splan == 4) hipLaunchKernelGGL(stft_generic_tiled_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, st, G);
    else if (splan == 5) hipLaunchKernelGGL(stft_generic_tiled_kernel<5>, dim3((unsigned)blocks), dim3(256), lds, st, G);
    else hipLaunchKernelGGL(stft_generic_tiled_kernel<0>, dim3((unsigned)blocks), dim3(256), lds, st, G);
    AT_LAUNCH_CHECK();
    return AT_OK;
  }
  if (mel_out != nullptr) return AT_ERR_UNSUPPORTED;
  GenArgs A;
  A.npass = generic_fft_plan(n_fft, A.radix);
  if (A.npass == 0) return AT_ERR_UNSUPPORTED;
  A.x = x; A.window = window; A.tw = reinterpret_cast<const float2*>(twiddles); A.out = reinterpret_cast<float2*>(stft_out);
  A.T = T; A.T2 = T + 2 * (int64_t)pad + right_pad; A.rows = rows; A.n_out = n_frames_out;
  A.frame_lo = frame_lo; A.hop = hop; A.pad = pad; A.pad_mode = pad_mode; A.M = n_fft / 2;
  const size_t lds = (size_t)n_fft * sizeof(float2);        // two buffers of M complex
  int e = allow_big_lds(reinterpret_cast<const void*>(stft_generic_kernel));
  if (e != AT_OK) return e;
  hipLaunchKernelGGL(stft_generic_kernel, dim3((unsigned)grid_for(rows * n_frames_out)), dim3(256), lds, st, A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

bool istft_tiled_supported(int n_fft, int hop) { return (n_fft == 4096 || n_fft == 8192) && hop * 4 == n_fft; }

int64_t istft_tiled_workspace_floats(int64_t n_frames, int n_fft, int hop) {
  return (n_frames - 1) * hop + n_fft + INV2_DUMP_FLOATS + 2 * (int64_t)(n_fft / 2 + 1) + 2;
}

int istft_tiled(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft, int hop,
                int lead, int64_t n_frames, int64_t length, float* out, float* workspace, hipStream_t st) {
  if (!istft_tiled_supported(n_fft, hop)) return AT_ERR_UNSUPPORTED;
  const int M = n_fft / 2, plan = M == 2048 ? 1 : 2, FB = plan == 1 ? 2 : 1;
  const int64_t env_n = (n_frames - 1) * hop + n_fft;           // even
  {
    int64_t eb = (env_n + 255) / 256;
    if (eb > 4096) eb = 4096;
    hipLaunchKernelGGL(istft_env_generic_kernel, dim3((unsigned)eb), dim3(256), 0, st, window, workspace, (int)n_frames, n_fft,
                       hop, env_n, (int64_t)INV2_DUMP_FLOATS + 2 * (M + 1) + 2);
    AT_LAUNCH_CHECK();
  }
  GenInv2Args G;
  G.X = reinterpret_cast<const float2*>(X); G.window = window; G.tw = reinterpret_cast<const float2*>(twiddles);
  G.inv_env = workspace; G.dump = workspace + env_n; G.zeros = reinterpret_cast<const float2*>(workspace + env_n + INV2_DUMP_FLOATS);
  G.out = out; G.rows = rows; G.length = length; G.n_x = (int)n_x; G.lead = lead; G.n_frames = (int)n_frames;
  G.n_seg = (int)n_frames - 1 + 4;
  G.run = INV2_RUN;
  G.runs_per_row = (G.n_seg + G.run - 1) / G.run;
  G.total_units = rows * G.runs_per_row;
  (void)FB;
  const size_t lds = ((size_t)4096 + 16 + (plan == 1 ? INV2_TW_SLOTS(1) : INV2_TW_SLOTS(2)) + (size_t)M) * sizeof(float2) + 64;
  const void* kfn = plan == 1 ? reinterpret_cast<const void*>(istft_tiled_pow2_kernel<1>) : reinterpret_cast<const void*>(istft_tiled_pow2_kernel<2>);
  int e = allow_big_lds(kfn);
  if (e != AT_OK) return e;
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu > 2 ? 2 : (per_cu < 1 ? 1 : per_cu);
  int64_t blocks = (int64_t)device_cu_count() * per_cu;
  if (blocks > G.total_units) blocks = G.total_units;
  if (plan == 1) hipLaunchKernelGGL(istft_tiled_pow2_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, G);
  else hipLaunchKernelGGL(istft_tiled_pow2_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, st, G);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// One-pass inverse of the run-time sizes (istft_generic_ola_kernel): frames per tile and history depth for (n_fft, hop),
// false when the shape keeps the frame buffer + gather path (odd hop or n_fft / 2, more history than a tile holds, LDS).
static bool generic_ola_plan(int n_fft, int hop, int* FBo, int* Ro) {
  static const int off = env_int_once("AT_ISTFT_GENERIC_OLA", 1) == 0;     // A/B: 0 = frame buffer + gather
  if (off || n_fft < 8 || (n_fft & 3) || (hop & 1) || hop <= 0 || hop > n_fft) return false;
  const int M = n_fft / 2;
  gfft::PassList pl;
  if (M > gfft::TILE_POINTS || !gfft::factor(M, &pl)) return false;
  const int R = (n_fft + hop - 1) / hop - 1;
  int fb = gfft::TILE_POINTS / M;
  if (fb > 64) fb = 64;
  // two workgroups per CU where the tile allows it
  while (fb > R && fb > 2 && (size_t)(R + fb + 3) * M * sizeof(float2) > 80 * 1024) --fb;
  if (fb < R || fb < 1 || (size_t)(R + fb + 3) * M * sizeof(float2) > 160 * 1024) return false;
  if (R > 16) return false;
  *FBo = fb; *Ro = R;
  return true;
}

bool istft_generic_ola_supported(int n_fft, int hop) {
  int fb, r;
  return generic_ola_plan(n_fft, hop, &fb, &r);
}

int64_t istft_generic_ola_workspace_floats(int64_t n_frames, int n_fft, int hop) { return (n_frames - 1) * hop + n_fft; }

int istft_generic_ola(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles, int n_fft,
                      int hop, int64_t length, float* out, float* workspace, hipStream_t st) {
  GenOlaArgs G;
  if (!generic_ola_plan(n_fft, hop, &G.FB, &G.R)) return AT_ERR_UNSUPPORTED;
  const int M = n_fft / 2;
  if (!gfft::factor(M, &G.pl)) return AT_ERR_UNSUPPORTED;
  const int64_t env_n = (n_frames - 1) * hop + n_fft;
  {
    int64_t eb = (env_n + 255) / 256;
    if (eb > 4096) eb = 4096;
    hipLaunchKernelGGL(istft_env_generic_kernel, dim3((unsigned)eb), dim3(256), 0, st, window, workspace, (int)n_frames, n_fft,
                       hop, env_n, (int64_t)0);
    AT_LAUNCH_CHECK();
  }
  G.X = reinterpret_cast<const float2*>(X); G.window = window; G.tw = reinterpret_cast<const float2*>(twiddles);
  G.inv_env = workspace; G.out = out; G.rows = rows; G.length = length; G.env_n = env_n;
  G.n_frames = (int)n_frames; G.M = M; G.h2 = hop / 2;
  static const int small_plans = env_int_once("AT_STFT_GENERIC_PLANS", 1);
  const int splan = !small_plans ? 0 : (M == 200 && G.FB == 20) ? 3 : (M == 600 && G.FB == 6) ? 4 : (M == 960 && G.FB == 4) ? 5 : 0;
  // pairs to produce per row: every output sample p = 2 j - M < length
  const int64_t need_pairs = (length + M + 1) / 2;
  const int64_t per_tile = (int64_t)G.FB * G.h2;
  G.tiles_per_row = (int)((need_pairs + per_tile - 1) / per_tile);
  G.vec2 = ((length & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0) ? 1 : 0;
  const size_t lds = (size_t)(G.R + G.FB + 3) * M * sizeof(float2);
  const void* kfn = splan == 3 ? reinterpret_cast<const void*>(istft_generic_ola_kernel<3>)
                  : splan == 4 ? reinterpret_cast<const void*>(istft_generic_ola_kernel<4>)
                  : splan == 5 ? reinterpret_cast<const void*>(istft_generic_ola_kernel<5>)
                               : reinterpret_cast<const void*>(istft_generic_ola_kernel<0>);
  int e = allow_big_lds(kfn);
  if (e != AT_OK) return e;
  int per_cu = (int)((160 * 1024) / lds);
  static const int cap_env = env_int_once("AT_ISTFT_GENERIC_WGS", 0);
  const int cap = cap_env > 0 ? cap_env : 3;
  per_cu = per_cu > cap ? cap : (per_cu < 1 ? 1 : per_cu);
  const int64_t slots = (int64_t)device_cu_count() * per_cu;
  // runs: >= 4 per workgroup slot for balance, >= 8 tiles each (a run that starts inside a row pays one extra tile)
  int64_t k = (4 * slots + rows - 1) / rows;
  const int64_t kmax = G.tiles_per_row / 8 > 0 ? G.tiles_per_row / 8 : 1;
  if (k > kmax) k = kmax;
  if (k < 1) k = 1;
  G.tiles_per_run = (int)((G.tiles_per_row + k - 1) / k);
  G.runs_per_row = (G.tiles_per_row + G.tiles_per_run - 1) / G.tiles_per_run;
  G.total_runs = rows * G.runs_per_row;
  int64_t blocks = slots < G.total_runs ? slots : G.total_runs;
  if (splan == 3) hipLaunchKernelGGL(istft_generic_ola_kernel<3>, dim3((unsigned)blocks), dim3(256), lds, st, G);
  else if (splan == 4) hipLaunchKernelGGL(istft_generic_ola_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, st, G);
  else if (splan == 5) hipLaunchKernelGGL(istft_generic_ola_kernel<5>, dim3((unsigned)blocks), dim3(256), lds, st, G);
  else hipLaunchKernelGGL(istft_generic_ola_kernel<0>, dim3((unsigned)blocks), dim3(256), lds, st, G);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

int istft_frames_generic(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                         int n_fft, float* frames, hipStream_t st) {
  const int Mh = n_fft / 2;
  const int64_t total_frames = rows * n_frames;
  static const int tiled_off = env_int_once("AT_ISTFT_GENERIC_OLD", 0);    // A/B: the one-frame-per-workgroup kernel
  if (!tiled_off && Mh <= gfft::TILE_POINTS && total_frames > 0) {
    GenInvTileArgs G;
    if (!gfft::factor(Mh, &G.pl)) return AT_ERR_UNSUPPORTED;
    int fb = gfft::TILE_POINTS / Mh;
    if (fb > 64) fb = 64;
    if (fb > total_frames) fb = (int)total_frames;
    G.X = reinterpret_cast<const float2*>(X); G.window = window; G.tw = reinterpret_cast<const float2*>(twiddles);
    G.frames = frames; G.total = total_frames; G.M = Mh; G.FB = fb;
    static const int small_plans = env_int_once("AT_STFT_GENERIC_PLANS", 1);
    const int splan = !small_plans ? 0 : (Mh == 200 && fb == 20) ? 3 : (Mh == 600 && fb == 6) ? 4 : (Mh == 960 && fb == 4) ? 5 : 0;
    const void* kfn = splan == 3 ? reinterpret_cast<const void*>(istft_frames_generic_tiled_kernel<3>)
                    : splan == 4 ? reinterpret_cast<const void*>(istft_frames_generic_tiled_kernel<4>)
                    : splan == 5 ? reinterpret_cast<const void*>(istft_frames_generic_tiled_kernel<5>)
                                 : reinterpret_cast<const void*>(istft_frames_generic_tiled_kernel<0>);
    const size_t lds_t = ((size_t)fb * Mh + 3 * (size_t)Mh) * sizeof(float2);
    int e = allow_big_lds(kfn);
    if (e != AT_OK) return e;
    int per_cu = (int)((160 * 1024) / lds_t);
    // the fixed plans need 74-90 registers (5-6 waves per SIMD): four workgroups where the LDS allows; the run-time plan 168
    static const int cap_env = env_int_once("AT_ISTFT_GENERIC_WGS", 0);      // A/B
    const int cap = cap_env > 0 ? cap_env : (splan ? 4 : 3);
    per_cu = per_cu > cap ? cap : (per_cu < 1 ? 1 : per_cu);
    const int64_t n_tiles = (total_frames + fb - 1) / fb;
    int64_t blocks = (int64_t)device_cu_count() * per_cu;
    if (blocks > n_tiles) blocks = n_tiles;
    if (splan == 3) hipLaunchKernelGGL(istft_frames_generic_tiled_kernel<3>, dim3((unsigned)blocks), dim3(256), lds_t, st, G);
    else if (splan == 4) hipLaunchKernelGGL(istft_frames_generic_tiled_kernel<4>, dim3((unsigned)blocks), dim3(256), lds_t, st, G);
    else if (splan == 5) hipLaunchKernelGGL(istft_frames_generic_tiled_kernel<5>, dim3((unsigned)blocks), dim3(256), lds_t, st, G);
    else hipLaunchKernelGGL(istft_frames_generic_tiled_kernel<0>, dim3((unsigned)blocks), dim3(256), lds_t, st, G);
    AT_LAUNCH_CHECK();
    return AT_OK;
  }
  GenInvArgs A;
  A.npass = generic_fft_plan(n_fft, A.radix);
  if (A.npass == 0) return AT_ERR_UNSUPPORTED;
  A.X = reinterpret_cast<const float2*>(X); A.window = window; A.tw = reinterpret_cast<const float2*>(twiddles);
  A.frames = frames; A.rows = rows; A.n_frames = n_frames; A.M = n_fft / 2;
  const size_t lds = (size_t)n_fft * sizeof(float2);
  int e = allow_big_lds(reinterpret_cast<const void*>(istft_frames_generic_kernel));
  if (e != AT_OK) return e;
  hipLaunchKernelGGL(istft_frames_generic_kernel, dim3((unsigned)grid_for(rows * n_frames)), dim3(256), lds, st, A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // namespace at
