// Inverse STFT for gfx950 (power-of-two n_fft in [32, 2048]).
//
// Replaces reference audiotools/core/audio_signal.py:1283-1290  torch.istft(X, n_fft, hop,
// window, length, center=True): per frame a C2R FFT, multiplication by the window, overlap-add,
// division by the overlap-added squared window, removal of the n_fft/2 centre padding.
//
// Kernel A (istft_frames_kernel): one wave per 64/L frames.  The Hermitian half spectrum of a
// frame is folded into the M-point complex spectrum of z[n] = x[2n] + i x[2n+1]
//     Z[k] = (X[k] + conj X[M-k]) + i e^{+2 pi i k/N} (X[k] - conj X[M-k])
// and z = conj(FFT_M(conj Z)) / N, i.e. the forward wave FFT of fft_wave.h is reused unchanged.
// The windowed frame goes to a (rows, frames, n_fft) float buffer with coalesced float2 stores.
// Kernel B (istft_ola_kernel): every output sample gathers its <= n_fft/hop frames and divides
// by the window^2 envelope (samples no frame covers, or with a vanishing envelope, are 0 like
// torch's zero padding to `length`).
#include "at_common.h"
#include "fft_wave.h"

namespace {

struct IstftArgs {
  const float2* X;       // (rows, n_frames, M+1) bin-contiguous
  const float* window;   // (N)
  const float2* tw;      // (N): (cos, -sin)(2 pi k / N)
  float* frames;         // (rows, n_frames, N)
  int64_t rows;
  int n_frames;
  int groups_per_row;
  int64_t total_groups;
};

template <int M>
__global__ __launch_bounds__(256) void istft_frames_kernel(const IstftArgs A) {
  using P = Plan<M>;
  constexpr int L = P::L, FW = P::FW, N = 2 * M;
  __shared__ float2 lds[4 * WAVE_LDS_SLOTS];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int fs = lane / L, t = lane % L;
  float2* wbuf = lds + wave * WAVE_LDS_SLOTS;
  float2* fbuf = wbuf + fs * P::SLOTS;
  constexpr int NB2 = 16 / P::R2;
  constexpr int NB3 = 16 / P::R3;

  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < A.total_groups; g += (int64_t)gridDim.x * 4) {
    const int64_t row = g / A.groups_per_row;
    const int gb = (int)(g - row * A.groups_per_row);
    const int f = gb * FW + fs;
    const bool live = f < A.n_frames;
    const float2* __restrict__ Xf = A.X + ((int64_t)row * A.n_frames + (live ? f : 0)) * (M + 1);

    // ---- fold the half spectrum: a[q] = conj(Z[k]), k = t + L q
    float2 a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = t + L * q;
      float2 xa = live ? Xf[k] : make_float2(0.f, 0.f);
      float2 xm = live ? Xf[M - k] : make_float2(0.f, 0.f);
      if (k == 0) { xa.y = 0.f; xm.y = 0.f; }  // c2r ignores the imaginary part of DC and Nyquist
      const float2 w = A.tw[k];                // (cos, -sin)(2 pi k / N)
      const float c = w.x, s = -w.y;
      const float sr = xa.x + xm.x, si = xa.y - xm.y;   // X[k] + conj X[M-k]
      const float dr = xa.x - xm.x, di = xa.y + xm.y;   // X[k] - conj X[M-k]
      const float zr = sr - s * dr - c * di;
      const float zi = si + c * dr - s * di;
      a[q] = make_float2(zr, -zi);
    }
    // ---- forward FFT of conj(Z)
    pass_compute_store<16, 1, L>(a, fbuf, t, nullptr);
    wave_sync();
    if constexpr (P::R2 > 1) {
      load_points<L>(a, fbuf, t);
      wave_sync();
      float2 tw2[NB2 * P::R2];
#pragma unroll
      for (int b = 0; b < NB2; ++b) {
        const int j = t + b * L;
#pragma unroll
        for (int r = 1; r < P::R2; ++r) tw2[b * P::R2 + r] = A.tw[r * (j % 16) * (N / (16 * P::R2))];
      }
      pass_compute_store<P::R2, 16, L>(a, fbuf, t, tw2);
      wave_sync();
    }
    if constexpr (P::R3 > 1) {
      load_points<L>(a, fbuf, t);
      wave_sync();
      constexpr int NS = 16 * P::R2;
      float2 tw3[NB3 * P::R3];
#pragma unroll
      for (int b = 0; b < NB3; ++b) {
        const int j = t + b * L;
#pragma unroll
        for (int r = 1; r < P::R3; ++r) tw3[b * P::R3 + r] = A.tw[r * (j % NS) * (N / (NS * P::R3))];
      }
      pass_compute_store<P::R3, NS, L>(a, fbuf, t, tw3);
      wave_sync();
    }
    // ---- z[n] = conj(Y[n]) / N, window, store
    const float inv_n = 1.0f / (float)N;
    float2* __restrict__ out = reinterpret_cast<float2*>(A.frames + ((int64_t)row * A.n_frames + f) * N);
    const float2* __restrict__ w2 = reinterpret_cast<const float2*>(A.window);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = t + L * q;
      const float2 y = fbuf[phys<L>(n)];
      const float2 w = w2[n];
      if (live) out[n] = make_float2(y.x * inv_n * w.x, -y.y * inv_n * w.y);
    }
    wave_sync();
  }
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        float* __restrict__ out, int64_t rows, int n_frames, int N,
                                                        int hop, int64_t length) {
  const int64_t total = rows * length;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / length;
    const int64_t p = i - row * length;
    const int64_t pp = p + N / 2;  // position in the centre-padded signal
    int64_t f_hi = pp / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    int64_t f_lo = (pp - N + hop) / hop;  // smallest f with f*hop + N > pp
    if (pp - N + 1 <= 0) f_lo = 0;
    if (f_lo < 0) f_lo = 0;
    float acc = 0.f, env = 0.f;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
      const int n = (int)(pp - f * hop);
      if (n < 0 || n >= N) continue;
      const float w = window[n];
      acc += frames[(row * n_frames + f) * (int64_t)N + n];
      env = fmaf(w, w, env);
    }
    out[i] = env > 1e-11f ? acc / env : 0.f;
  }
}

template <int M>
int launch_frames(const IstftArgs& A, hipStream_t stream) {
  int64_t blocks = (A.total_groups + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(istft_frames_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, stream, A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // namespace

extern "C" {

// bytes of the (rows, n_frames, n_fft) float frame buffer
int64_t at_istft_workspace_bytes(int64_t rows, int64_t n_frames, int n_fft) {
  if (rows < 0 || n_frames < 0 || n_fft <= 0) return AT_ERR_INVALID;
  return rows * n_frames * (int64_t)n_fft * 4;
}

// X (rows, n_frames, n_fft/2+1) complex64 interleaved, bin-contiguous (as at_stft_mel_f32 writes it);
// out (rows, length): sample p comes from centre-padded position p + n_fft/2.
int at_istft_f32(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles, int n_fft,
                 int hop, int64_t length, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!X || !window || !twiddles || !out || rows < 0 || n_frames <= 0 || hop <= 0 || length < 0) return AT_ERR_INVALID;
  if (!(n_fft >= 32 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0)) return AT_ERR_UNSUPPORTED;
  if (n_frames >= (1LL << 31)) return AT_ERR_UNSUPPORTED;
  if (rows == 0 || length == 0) return AT_OK;
  if (!workspace || workspace_bytes < at_istft_workspace_bytes(rows, n_frames, n_fft)) return AT_ERR_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int M = n_fft / 2;
  const int FW = 64 / (M / 16);
  IstftArgs A;
  A.X = reinterpret_cast<const float2*>(X); A.window = window; A.tw = reinterpret_cast<const float2*>(twiddles);
  A.frames = reinterpret_cast<float*>(workspace); A.rows = rows; A.n_frames = (int)n_frames;
  A.groups_per_row = (int)((n_frames + FW - 1) / FW);
  A.total_groups = rows * A.groups_per_row;
  int rc = AT_ERR_UNSUPPORTED;
  switch (M) {
    case 16: rc = launch_frames<16>(A, st); break;
    case 32: rc = launch_frames<32>(A, st); break;
    case 64: rc = launch_frames<64>(A, st); break;
    case 128: rc = launch_frames<128>(A, st); break;
    case 256: rc = launch_frames<256>(A, st); break;
    case 512: rc = launch_frames<512>(A, st); break;
    case 1024: rc = launch_frames<1024>(A, st); break;
  }
  if (rc != AT_OK) return rc;
  const int64_t total = rows * length;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A.frames, window, out, rows,
                     (int)n_frames, n_fft, hop, length);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // extern "C"
