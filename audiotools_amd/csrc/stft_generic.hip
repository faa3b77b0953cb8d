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
    if (R == 4) stockham_pass<4>(a, b, tw, M, NS);
    else if (R == 2) stockham_pass<2>(a, b, tw, M, NS);
    else if (R == 3) stockham_pass<3>(a, b, tw, M, NS);
    else if (R == 5) stockham_pass<5>(a, b, tw, M, NS);
    else stockham_pass<7>(a, b, tw, M, NS);
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
  while (m % 4 == 0 && n < 16) { radix[n++] = 4; m /= 4; }
  while (m % 2 == 0 && n < 16) { radix[n++] = 2; m /= 2; }
  while (m % 3 == 0 && n < 16) { radix[n++] = 3; m /= 3; }
  while (m % 5 == 0 && n < 16) { radix[n++] = 5; m /= 5; }
  while (m % 7 == 0 && n < 16) { radix[n++] = 7; m /= 7; }
  if (m != 1) return 0;
  if (n == 0) { radix[0] = 1; return 0; }     // n_fft == 2: not worth a kernel
  return n;
}

static int grid_for(int64_t frames) {
  const int64_t cap = (int64_t)device_cu_count() * 8;
  return (int)(frames < cap ? frames : cap);
}

int stft_generic(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles, int n_fft, int hop,
                 int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out, float* stft_out, hipStream_t st) {
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
