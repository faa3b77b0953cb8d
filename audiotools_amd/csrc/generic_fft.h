// Mixed-radix (2, 3, 4, 5, 7) workgroup FFT used for the transform sizes the wave-FFT kernels do not
// cover (n_fft = 4096 ... 16384, and even non-power-of-two windows such as 400 / 1200 / 1920).
#pragma once
#include "at_common.h"

namespace at {

// radix butterflies shared by the workgroup FFT kernels (stft_generic.hip, longconv.hip)
namespace gfft {
constexpr int MAX_RADIX = 7;

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// forward DFT of R points in place
template <int R>
__device__ __forceinline__ void dft_r(float2 (&v)[MAX_RADIX]);
template <>
__device__ __forceinline__ void dft_r<2>(float2 (&v)[MAX_RADIX]) {
  const float2 a = v[0], b = v[1];
  v[0] = make_float2(a.x + b.x, a.y + b.y);
  v[1] = make_float2(a.x - b.x, a.y - b.y);
}
template <>
__device__ __forceinline__ void dft_r<4>(float2 (&v)[MAX_RADIX]) {
  const float2 t0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
  const float2 t2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), t3 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
  v[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
  v[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
  v[1] = make_float2(t1.x + t3.y, t1.y - t3.x);   // t1 - i t3
  v[3] = make_float2(t1.x - t3.y, t1.y + t3.x);   // t1 + i t3
}
template <>
__device__ __forceinline__ void dft_r<3>(float2 (&v)[MAX_RADIX]) {
  const float S3 = 0.86602540378443864676f;       // sin(2 pi / 3)
  const float2 s = make_float2(v[1].x + v[2].x, v[1].y + v[2].y);
  const float2 d = make_float2(v[1].x - v[2].x, v[1].y - v[2].y);
  const float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
  v[0] = make_float2(v[0].x + s.x, v[0].y + s.y);
  // X1 = m - i S3 d,  X2 = m + i S3 d
  v[1] = make_float2(m.x + S3 * d.y, m.y - S3 * d.x);
  v[2] = make_float2(m.x - S3 * d.y, m.y + S3 * d.x);
}
template <>
__device__ __forceinline__ void dft_r<5>(float2 (&v)[MAX_RADIX]) {
  const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;   // cos(2 pi/5), cos(4 pi/5)
  const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;    // sin(2 pi/5), sin(4 pi/5)
  const float2 a1 = make_float2(v[1].x + v[4].x, v[1].y + v[4].y), b1 = make_float2(v[1].x - v[4].x, v[1].y - v[4].y);
  const float2 a2 = make_float2(v[2].x + v[3].x, v[2].y + v[3].y), b2 = make_float2(v[2].x - v[3].x, v[2].y - v[3].y);
  const float2 x0 = v[0];
  v[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
  const float2 p1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x, x0.y + C1 * a1.y + C2 * a2.y);
  const float2 p2 = make_float2(x0.x + C2 * a1.x + C1 * a2.x, x0.y + C2 * a1.y + C1 * a2.y);
  const float2 q1 = make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
  const float2 q2 = make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
  // X_k = p - i q  (k = 1, 2),  X_{5-k} = p + i q
  v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);
  v[4] = make_float2(p1.x - q1.y, p1.y + q1.x);
  v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);
  v[3] = make_float2(p2.x - q2.y, p2.y + q2.x);
}

template <>
__device__ __forceinline__ void dft_r<7>(float2 (&v)[MAX_RADIX]) {
  const float C1 = 0.62348980185873353053f, C2 = -0.22252093395631440429f, C3 = -0.90096886790241912624f;  // cos(2 pi m/7)
  const float S1 = 0.78183148246802980871f, S2 = 0.97492791218182360702f, S3 = 0.43388373911755812048f;   // sin(2 pi m/7)
  const float2 a1 = make_float2(v[1].x + v[6].x, v[1].y + v[6].y), b1 = make_float2(v[1].x - v[6].x, v[1].y - v[6].y);
  const float2 a2 = make_float2(v[2].x + v[5].x, v[2].y + v[5].y), b2 = make_float2(v[2].x - v[5].x, v[2].y - v[5].y);
  const float2 a3 = make_float2(v[3].x + v[4].x, v[3].y + v[4].y), b3 = make_float2(v[3].x - v[4].x, v[3].y - v[4].y);
  const float2 x0 = v[0];
  v[0] = make_float2(x0.x + a1.x + a2.x + a3.x, x0.y + a1.y + a2.y + a3.y);
  const float2 p1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x + C3 * a3.x, x0.y + C1 * a1.y + C2 * a2.y + C3 * a3.y);
  const float2 p2 = make_float2(x0.x + C2 * a1.x + C3 * a2.x + C1 * a3.x, x0.y + C2 * a1.y + C3 * a2.y + C1 * a3.y);
  const float2 p3 = make_float2(x0.x + C3 * a1.x + C1 * a2.x + C2 * a3.x, x0.y + C3 * a1.y + C1 * a2.y + C2 * a3.y);
  const float2 q1 = make_float2(S1 * b1.x + S2 * b2.x + S3 * b3.x, S1 * b1.y + S2 * b2.y + S3 * b3.y);
  const float2 q2 = make_float2(S2 * b1.x - S3 * b2.x - S1 * b3.x, S2 * b1.y - S3 * b2.y - S1 * b3.y);
  const float2 q3 = make_float2(S3 * b1.x - S1 * b2.x + S2 * b3.x, S3 * b1.y - S1 * b2.y + S2 * b3.y);
  // X_k = p_k - i q_k,  X_{7-k} = p_k + i q_k
  v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);
  v[6] = make_float2(p1.x - q1.y, p1.y + q1.x);
  v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);
  v[5] = make_float2(p2.x - q2.y, p2.y + q2.x);
  v[3] = make_float2(p3.x + q3.y, p3.y - q3.x);
  v[4] = make_float2(p3.x - q3.y, p3.y + q3.x);
}
}  // namespace gfft

// M = n_fft / 2 factors into {2, 3, 5, 7}: fills radix[] (4s first) and returns the pass count, 0 if not.
int generic_fft_plan(int n_fft, int* radix /* [16] */);

// STFT of `n_frames_out` frames per row (same argument meaning as at_stft_mel_f32, no mel).
int stft_generic(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles, int n_fft, int hop,
                 int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out, float* stft_out, hipStream_t st);

// Inverse transform of every frame: X (rows, n_frames, n_fft/2+1) -> windowed frames (rows, n_frames, n_fft),
// the input of istft_ola_kernel.
int istft_frames_generic(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                         int n_fft, float* frames, hipStream_t st);

}  // namespace at
