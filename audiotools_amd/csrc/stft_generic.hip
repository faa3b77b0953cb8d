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
//    passes for 2048 / 4096 points instead of six;
//  * persistent workgroups walk RUNS of consecutive frames of one row inside their XCD's span of the
//    (row, frame) space, so the 4x overlap between neighbouring frames is served by L1 / that XCD's
//    L2 and HBM sees every sample once (the round-2 kernel dealt frames round-robin over the XCDs:
//    each of them re-read the overlap, input traffic = output traffic);
//  * the samples of the next tile are loaded into registers before the passes of the current one;
//  * every bin X[k], k = 0..M, comes from ONE formula on (Z[k mod M], Z[(M - k) mod M], w_N^k) -- DC and
//    Nyquist included -- in a single ascending sweep: 512-byte store segments, no special cases;
//  * optional fused mel epilogue (audio_signal.py:1355-1368): |X| goes to LDS (over the transform
//    buffer) and every (frame, band) pair is a 16-lane dot product over the band's non-zero span
//    (banded tables: {first bin, length, weight offset} per band) -- no second pass over stft_data,
//    no dense matmul.
struct Gen2Args {
  const float* x;
  const float* window;
  const float2* tw;        // (N): (cos, -sin)(2 pi k / N)
  float2* out;             // (rows, n_out, M + 1)
  float* mel;              // (rows, n_out, n_mels) or null
  const int* band;         // (n_mels, 3): first bin, length, offset into bw
  const float* bw;         // concatenated non-zero spans of the filterbank rows
  int64_t T, T2, rows, n_out;
  int64_t tiles_per_row, total_tiles;
  int frame_lo, hop, pad, pad_mode;
  int M, FB, n_mels, run, n_xcd, vec2;
  at::gfft::PassList pl;
};

constexpr int G2_LOADS = at::gfft::TILE_POINTS / 256;     // float2 samples of a tile per thread

__global__ __launch_bounds__(256, 3) void stft_generic_tiled_kernel(const Gen2Args A) {
  extern __shared__ __attribute__((aligned(16))) float2 gbuf[];
  const int M = A.M, N = 2 * M, FB = A.FB;
  float2* buf = gbuf;                 // [FB][M]
  float2* tw = gbuf + FB * M;         // [M]: w_M^t
  float* mag = reinterpret_cast<float*>(gbuf);   // [FB][M + 1], over the transform buffer once Z is consumed
  for (int i = threadIdx.x; i < M; i += 256) tw[i] = A.tw[2 * i];
  const at::gfft::RowLayout lay{M};
  const int npts = FB * M;

  // persistent schedule: one contiguous span of tiles per XCD, runs of A.run tiles per workgroup
  const int n_x = (int)gridDim.x < A.n_xcd ? (int)gridDim.x : A.n_xcd;
  const int xcd = blockIdx.x % n_x, lblk = blockIdx.x / n_x;
  const int nblk_x = ((int)gridDim.x - xcd + n_x - 1) / n_x;
  const int64_t g_lo = A.total_tiles * xcd / n_x, g_hi = A.total_tiles * (xcd + 1) / n_x;

  float2 r[G2_LOADS];
  const int Ti = (int)A.T, n_out = (int)A.n_out, tpr = (int)A.tiles_per_row, hop = A.hop;
  // tile g = (row, t): frames f0 = t FB .. f0 + FB - 1; s0 = first sample of frame f0 (outer-padded coordinates)
  auto interior = [&](int f0, int s0) {
    return A.vec2 && A.pad == 0 && s0 >= 0 && s0 + (FB - 1) * hop + N <= Ti && f0 + FB <= n_out;
  };
  auto fetch = [&](const float* __restrict__ xs, int tid) __attribute__((always_inline)) {     // xs = x row + s0
#pragma unroll
    for (int i = 0; i < G2_LOADS; ++i) {
      if (256 * i < npts) {                               // uniform; lanes past the end repeat the last point
        const int e = min(tid + 256 * i, npts - 1);
        const int fi = e >= M ? 1 : 0, n = e - fi * M;    // FB <= 2
        r[i] = *reinterpret_cast<const float2*>(xs + fi * hop + 2 * n);
      }
    }
  };
  constexpr int KB = at::gfft::TILE_POINTS / 256 + 1;     // bins of a tile per thread (FB (M + 1) <= 4098)
  const int nbins = FB * (M + 1);

  for (int64_t gbase = g_lo + (int64_t)lblk * A.run; gbase < g_hi; gbase += (int64_t)nblk_x * A.run) {
    const int64_t g_end = min(gbase + A.run, g_hi);
    bool have = false;
    for (int64_t g = gbase; g < g_end; ++g) {
      // opaque copy of the thread index per tile: nothing indexed by it is hoisted out of the persistent loops
      // (the 17 unrolled bin slots and every pass variant would otherwise keep their addresses live: 800 B of spills)
      int tid = (int)threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int row = (int)(g / tpr);
      const int f0 = (int)(g - (int64_t)row * tpr) * FB;
      const int s0 = (f0 + A.frame_lo) * hop - M;
      const float* __restrict__ xr = A.x + (int64_t)row * A.T;
      const bool fast = interior(f0, s0);
      if (fast && !have) fetch(xr + s0, tid);
      // ---- windowed samples -> LDS
      if (fast) {
        const float2* __restrict__ w2 = reinterpret_cast<const float2*>(A.window);
#pragma unroll
        for (int i = 0; i < G2_LOADS; ++i) {
          if (256 * i < npts) {
            const int e = min(tid + 256 * i, npts - 1);
            const int n = e >= M ? e - M : e;
            const float2 w = w2[n];
            buf[e] = make_float2(r[i].x * w.x, r[i].y * w.y);
          }
        }
      } else {
#pragma unroll 1
        for (int e = tid; e < npts; e += 256) {
          const int fi = e >= M ? 1 : 0, n = e - fi * M;
          float a = 0.f, b = 0.f;
          if (f0 + fi < n_out) {
            const int64_t sidx = (int64_t)s0 + fi * hop + 2 * n;
            a = at::fetch_padded(xr, sidx, A.T, A.T2, A.pad, A.pad_mode);
            b = at::fetch_padded(xr, sidx + 1, A.T, A.T2, A.pad, A.pad_mode);
          }
          buf[e] = make_float2(a * A.window[2 * n], b * A.window[2 * n + 1]);
        }
      }
      __syncthreads();
      // ---- the next tile's samples: issued now, consumed after this tile's stores
      have = false;
      if (g + 1 < g_end) {
        const int row2 = (int)((g + 1) / tpr);
        const int f02 = (int)(g + 1 - (int64_t)row2 * tpr) * FB;
        const int s02 = (f02 + A.frame_lo) * hop - M;
        if (interior(f02, s02)) { fetch(A.x + (int64_t)row2 * A.T + s02, tid); have = true; }
      }
      at::gfft::run_passes<256>(buf, tw, M, A.pl, FB, lay);
      // ---- split step, one ascending sweep over k = 0 .. M per frame; |X| kept for the mel stage
      float mg[KB];
      float2* __restrict__ orow = A.out + ((int64_t)row * n_out + f0) * (M + 1);
#pragma unroll
      for (int it = 0; it < KB; ++it) {
        const int e = tid + 256 * it;
        mg[it] = 0.f;
        if (e < nbins) {
          const int fi = e >= M + 1 ? 1 : 0, k = e - fi * (M + 1);
          const float2 zk = buf[fi * M + (k == M ? 0 : k)], zm = buf[fi * M + ((k == 0 || k == M) ? 0 : M - k)];
          const float2 w = A.tw[k];
          const float c = w.x, sn = -w.y;
          const float sr = zk.x + zm.x, si = zk.y - zm.y;
          const float dr = zk.x - zm.x, di = zk.y + zm.y;
          const float pp = fmaf(sn, dr, -c * di);
          const float qq = fmaf(sn, di, c * dr);
          float2 X = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));
          if (k == 0 || k == M) X.y = 0.f;                   // exactly real
          if (f0 + fi < n_out) orow[e] = X;                  // frame f0 + 1 follows frame f0: e indexes both rows
          mg[it] = __builtin_amdgcn_sqrtf(fmaf(X.x, X.x, X.y * X.y));
        }
      }
      if (A.mel != nullptr) {
        __syncthreads();                                      // every Z has been read: the buffer becomes |X|
#pragma unroll
        for (int it = 0; it < KB; ++it) {
          const int e = tid + 256 * it;
          if (e < nbins) mag[e] = mg[it];
        }
        __syncthreads();
        const int grp = tid >> 4, l16 = tid & 15;
        float* __restrict__ mel0 = A.mel + ((int64_t)row * n_out + f0) * A.n_mels;
#pragma unroll 1
        for (int task = grp; task < FB * A.n_mels; task += 16) {
          const int fi = task >= A.n_mels ? 1 : 0, m = task - fi * A.n_mels;
          const int lo = A.band[3 * m], len = A.band[3 * m + 1], off = A.band[3 * m + 2];
          const float* __restrict__ mrow = mag + fi * (M + 1) + lo;
          const float* __restrict__ wrow = A.bw + off;
          float acc = 0.f;
          for (int j = l16; j < len; j += 16) acc = fmaf(wrow[j], mrow[j], acc);
          acc += __shfl_xor(acc, 8, 16);
          acc += __shfl_xor(acc, 4, 16);
          acc += __shfl_xor(acc, 2, 16);
          acc += __shfl_xor(acc, 1, 16);
          if (l16 == 0 && f0 + fi < n_out) mel0[task] = acc;      // (frame fi, band m) -> fi * n_mels + m = task
        }
      }
      __syncthreads();                                        // the buffer is free for the next tile
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
                 const int* mel_band, const float* mel_w, int n_mels, float* mel_out, hipStream_t st) {
  static const int force_old = env_int_once("AT_STFT_GENERIC_OLD", 0);    // A/B: the round-2 one-frame-per-workgroup kernel
  if (n_fft / 2 <= gfft::TILE_POINTS && !(force_old && mel_out == nullptr)) {
    Gen2Args G;
    if (!gfft::factor(n_fft / 2, &G.pl)) return AT_ERR_UNSUPPORTED;
    G.x = x; G.window = window; G.tw = reinterpret_cast<const float2*>(twiddles); G.out = reinterpret_cast<float2*>(stft_out);
    G.mel = mel_out; G.band = mel_band; G.bw = mel_w; G.n_mels = mel_out ? n_mels : 0;
    G.T = T; G.T2 = T + 2 * (int64_t)pad + right_pad; G.rows = rows; G.n_out = n_frames_out;
    G.frame_lo = frame_lo; G.hop = hop; G.pad = pad; G.pad_mode = pad_mode; G.M = n_fft / 2;
    G.FB = G.M <= gfft::TILE_POINTS / 2 ? 2 : 1;
    G.tiles_per_row = (n_frames_out + G.FB - 1) / G.FB;
    G.total_tiles = rows * G.tiles_per_row;
    G.vec2 = ((T % 2) == 0 && (hop % 2) == 0 && (G.M % 2) == 0 && (reinterpret_cast<uintptr_t>(x) % 8) == 0) ? 1 : 0;
    const size_t lds = ((size_t)G.FB * G.M + G.M) * sizeof(float2) + 16;     // + the Nyquist column of the |X| rows
    int e = allow_big_lds(reinterpret_cast<const void*>(stft_generic_tiled_kernel));
    if (e != AT_OK) return e;
    const int n_cu = device_cu_count();
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
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
    hipLaunchKernelGGL(stft_generic_tiled_kernel, dim3((unsigned)blocks), dim3(256), lds, st, G);
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

int istft_frames_generic(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                         int n_fft, float* frames, hipStream_t st) {
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
