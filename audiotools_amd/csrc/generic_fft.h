// Mixed-radix (2, 3, 4, 5) workgroup FFT used for the transform sizes the wave-FFT kernels do not
// cover (n_fft = 4096 ... 16384, and even non-power-of-two windows such as 400 / 1200 / 1920).
#pragma once
#include "at_common.h"

namespace at {

// M = n_fft / 2 factors into {2, 3, 5}: fills radix[] (4s first) and returns the pass count, 0 if not.
int generic_fft_plan(int n_fft, int* radix /* [16] */);

// STFT of `n_frames_out` frames per row (same argument meaning as at_stft_mel_f32, no mel).
int stft_generic(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles, int n_fft, int hop,
                 int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out, float* stft_out, hipStream_t st);

// Inverse transform of every frame: X (rows, n_frames, n_fft/2+1) -> windowed frames (rows, n_frames, n_fft),
// the input of istft_ola_kernel.
int istft_frames_generic(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                         int n_fft, float* frames, hipStream_t st);

}  // namespace at
