// Phase-vocoder time-scale modification of stft_data (gfx950).
//
// Replaces the CPU libsox round trips of reference audiotools/core/effects.py:247-309
//     time_stretch:  sox "tempo [-q] factor" + "rate"      pitch_shift: sox "pitch [-q] cents" + "rate"
// (torchaudio.sox_effects on .cpu() tensors, one setting per batch) by a device-side
// behavioural equivalent:  stft -> phase vocoder -> istft  (+ the polyphase resampler for pitch).
// sox's WSOLA output cannot be reproduced sample for sample ("parity
// unpinned"); the algorithm here is the textbook phase vocoder in the form torchaudio's
// functional.phase_vocoder states it, with the rate as an exact rational p/q so that frame
// positions are computed in integers (identical on host and device):
//     position of output frame k:  t_k = k p / q,  j = floor(t_k),  alpha = t_k - j
//     |Y_k| = (1 - alpha) |X_j| + alpha |X_{j+1}|                       (X_n = 0 for n >= N_in)
//     phase_k = angle X_0 + sum_{i<k} [ wrap(angle X_{j_i+1} - angle X_{j_i} - w_f hop) + w_f hop ],
//     w_f hop = pi hop f / (F - 1)   (used mod 2 pi)
// One thread owns one (row, bin) and walks the output frames in order (the phase is a running
// sum); the lanes of a wave are adjacent bins, so every load / store of a frame row is coalesced
// on the physical (rows, frames, bins) layout.  X_{j+1} of step k is X_j of a later step: the two
// frames are carried in registers, each input frame is read once.
#include "at_common.h"

namespace {

__global__ __launch_bounds__(256) void phase_vocoder_kernel(const float2* __restrict__ X, float2* __restrict__ Y,
                                                            int64_t rows, int n_in, int n_out, int F, int64_t p, int64_t q,
                                                            int hop) {
  const int64_t row = blockIdx.y;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const float2* __restrict__ Xr = X + row * (int64_t)n_in * F + f;
  float2* __restrict__ Yr = Y + row * (int64_t)n_out * F + f;
  const float TWO_PI = 6.283185307179586f;
  // expected phase advance per hop, 2 pi hop f / n_fft with n_fft = 2 (F - 1), reduced mod 2 pi in
  // INTEGER arithmetic: only its value mod 2 pi matters, and at full size (~1600 rad for the top
  // bin) float32 carries it with an error of 1e-4 rad per frame, which the running sum accumulates
  const int64_t n_fft = 2 * (int64_t)(F - 1);
  const float adv = (float)(6.283185307179586476925286766559 * (double)(((int64_t)hop * f) % n_fft) / (double)n_fft);
  auto load = [&](int64_t j) { return j < n_in ? Xr[j * F] : make_float2(0.f, 0.f); };
  int64_t j_cur = 0;
  float2 x0 = load(0), x1 = load(1);
  float a0 = atan2f(x0.y, x0.x), a1 = atan2f(x1.y, x1.x);
  float m0 = hypotf(x0.x, x0.y), m1 = hypotf(x1.x, x1.y);
  double acc = a0;    // phase_acc of output frame 0 = angle X_0; the running sum is kept in double
                      // (float32 round-off of ~900 additions would otherwise reach 5e-4 of the output)
  for (int k = 0; k < n_out; ++k) {
    const int64_t num = (int64_t)k * p;
    const int64_t j = num / q;
    const float alpha = (float)(num - j * q) / (float)q;
    while (j_cur < j) {   // advance the two-frame window (usually 0 or 1 steps; more when p/q > 1)
      ++j_cur;
      x0 = x1; a0 = a1; m0 = m1;
      x1 = load(j_cur + 1);
      a1 = atan2f(x1.y, x1.x);
      m1 = hypotf(x1.x, x1.y);
    }
    const float mag = alpha * m1 + (1.f - alpha) * m0;
    float sn, cs;
    {
      // reduce to [-pi, pi] in double before the float32 sincos
      const double TWO_PI_D = 6.283185307179586476925286766559;
      const float r = (float)(acc - TWO_PI_D * rint(acc / TWO_PI_D));
      sincosf(r, &sn, &cs);
    }
    Yr[(int64_t)k * F] = make_float2(mag * cs, mag * sn);
    float d = a1 - a0 - adv;
    d = d - TWO_PI * rintf(d / TWO_PI);
    acc += (double)(d + adv);
  }
}

}  // namespace

extern "C" {

// frames produced from n_in input frames at rate p/q: ceil(n_in q / p)
int64_t at_phase_vocoder_frames(int64_t n_in, int64_t p, int64_t q) {
  if (n_in <= 0 || p <= 0 || q <= 0) return AT_ERR_INVALID;
  return (n_in * q + p - 1) / p;
}

// X (rows, n_in, F) complex64 -> Y (rows, n_out, F) complex64, n_out = at_phase_vocoder_frames().
int at_phase_vocoder_f32(const float* X, int64_t rows, int64_t n_in, int64_t F, int64_t p, int64_t q, int hop,
                         float* Y, int64_t n_out, void* stream) {
  if (rows == 0) return AT_OK;
  if (!X || !Y || rows < 0 || n_in <= 0 || F < 2 || p <= 0 || q <= 0 || hop <= 0) return AT_ERR_INVALID;
  if (n_out != at_phase_vocoder_frames(n_in, p, q)) return AT_ERR_INVALID;
  if (n_in >= (1LL << 31) || n_out >= (1LL << 31) || F >= (1LL << 31) || rows > 65535LL * 65535LL) return AT_ERR_UNSUPPORTED;
  if ((__int128)n_out * p >= ((__int128)1 << 62)) return AT_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // rows ride on blockIdx.y in chunks of 65535
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid((unsigned)((F + 255) / 256), (unsigned)nr);
    hipLaunchKernelGGL(phase_vocoder_kernel, grid, dim3(256), 0, st,
                       reinterpret_cast<const float2*>(X) + r0 * n_in * F, reinterpret_cast<float2*>(Y) + r0 * n_out * F,
                       nr, (int)n_in, (int)n_out, (int)F, p, q, hop);
    AT_LAUNCH_CHECK();
  }
  return AT_OK;
}

}  // extern "C"
