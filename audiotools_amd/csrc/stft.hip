// Fused STFT (+ optional mel filterbank) for gfx950.
//
// Replaces, for power-of-two n_fft in [32, 2048]:
//   reference audiotools/core/audio_signal.py:1192-1202  F.pad + torch.stft(center=True)
//   reference audiotools/core/audio_signal.py:1355-1368  torch.abs + mel matmul
//
// One kernel does: (outer pad by index math) -> reflect centre-pad -> window
// -> n_fft-point real FFT -> write bin-contiguous (rows, frames, n_fft/2+1)
// complex64 -> |X| -> banded mel filterbank -> write (rows, frames, n_mels).
//
// Design (CDNA4):
//  * the real FFT of length N is a complex FFT of length M = N/2 on
//    z[n] = x[2n] + i x[2n+1] plus a split step;
//  * every thread owns 16 complex points; L = M/16 threads form one frame, so a
//    wave64 transforms 64/L frames at once (1 frame of n_fft=2048, 4 of 512);
//  * Stockham autosort passes of radix 16,16,{2,4} in registers; the exchange
//    between passes goes through a per-wave LDS slab of 8704 B (index i stored
//    at i + i/16 so the stride-16 writes of pass 1 spread over all banks).
//    Waves never synchronise with each other (no s_barrier in the kernel);
//  * a wave walks a chunk of consecutive frames of one row, so the 4x overlap
//    of the input between frames is served by L1/L2, HBM sees each sample once;
//  * twiddles and the window are loaded once per wave and stay in registers;
//  * global loads are float2 per lane, 512 B contiguous per wave instruction;
//    global stores are float2 per lane, 512 B contiguous;
//  * mel: the Slaney filterbank is banded (each bin feeds <= 2 bands), so it is
//    applied as "units" of 16 bins x 1 band from an LDS copy of |X| instead of a
//    dense (F x n_mels) GEMM: 2 flop/bin instead of 2*n_mels flop/bin.
#include "at_common.h"

namespace {

using at::cadd;
using at::cmul;
using at::csub;
using at::wave_sync;

// ---------------------------------------------------------------- small DFTs
// cos/sin(2*pi*k/16)
__device__ constexpr float C16[16] = {
    1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
    0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
    -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
    0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__device__ constexpr float S16[16] = {
    0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
    1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
    0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
    -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};

// a * exp(-2*pi*i*K/16), K compile-time
template <int K>
__device__ __forceinline__ float2 mul_w16(float2 a) {
  constexpr int k = ((K % 16) + 16) % 16;
  if constexpr (k == 0) return a;
  else if constexpr (k == 4) return make_float2(a.y, -a.x);
  else if constexpr (k == 8) return make_float2(-a.x, -a.y);
  else if constexpr (k == 12) return make_float2(-a.y, a.x);
  else {
    constexpr float c = C16[k], s = S16[k];  // w = c - i s
    return make_float2(fmaf(a.x, c, a.y * s), fmaf(a.y, c, -a.x * s));
  }
}

__device__ __forceinline__ void dft2(float2& a0, float2& a1) {
  float2 t = a0;
  a0 = cadd(t, a1);
  a1 = csub(t, a1);
}

// forward 4-point DFT, natural order in / natural order out
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  float2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
  float2 t2 = cadd(a1, a3), t3 = csub(a1, a3);
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = make_float2(t1.x + t3.y, t1.y - t3.x);  // t1 - i t3
  a3 = make_float2(t1.x - t3.y, t1.y + t3.x);  // t1 + i t3
}

template <int R>
struct Dft;

template <>
struct Dft<2> {
  static __device__ __forceinline__ void run(float2 (&v)[2]) { dft2(v[0], v[1]); }
};
template <>
struct Dft<4> {
  static __device__ __forceinline__ void run(float2 (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Dft<8> {
  // n = 2 n1 + n2, k = k1 + 4 k2
  static __device__ __forceinline__ void run(float2 (&v)[8]) {
    float2 e[4] = {v[0], v[2], v[4], v[6]};
    float2 o[4] = {v[1], v[3], v[5], v[7]};
    dft4(e[0], e[1], e[2], e[3]);
    dft4(o[0], o[1], o[2], o[3]);
    o[1] = mul_w16<2>(o[1]);
    o[2] = mul_w16<4>(o[2]);
    o[3] = mul_w16<6>(o[3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      v[k1] = cadd(e[k1], o[k1]);
      v[k1 + 4] = csub(e[k1], o[k1]);
    }
  }
};
template <>
struct Dft<16> {
  // n = 4 n1 + n2, k = k1 + 4 k2
  static __device__ __forceinline__ void run(float2 (&v)[16]) {
    float2 A0[4] = {v[0], v[4], v[8], v[12]};
    float2 A1[4] = {v[1], v[5], v[9], v[13]};
    float2 A2[4] = {v[2], v[6], v[10], v[14]};
    float2 A3[4] = {v[3], v[7], v[11], v[15]};
    dft4(A0[0], A0[1], A0[2], A0[3]);
    dft4(A1[0], A1[1], A1[2], A1[3]);
    dft4(A2[0], A2[1], A2[2], A2[3]);
    dft4(A3[0], A3[1], A3[2], A3[3]);
    // twiddle W16^(n2*k1)
    A1[1] = mul_w16<1>(A1[1]); A1[2] = mul_w16<2>(A1[2]); A1[3] = mul_w16<3>(A1[3]);
    A2[1] = mul_w16<2>(A2[1]); A2[2] = mul_w16<4>(A2[2]); A2[3] = mul_w16<6>(A2[3]);
    A3[1] = mul_w16<3>(A3[1]); A3[2] = mul_w16<6>(A3[2]); A3[3] = mul_w16<9>(A3[3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      float2 b0 = A0[k1], b1 = A1[k1], b2 = A2[k1], b3 = A3[k1];
      dft4(b0, b1, b2, b3);
      v[k1] = b0; v[k1 + 4] = b1; v[k1 + 8] = b2; v[k1 + 12] = b3;
    }
  }
};

// ------------------------------------------------------------------- planning
template <int M>
struct Plan {
  static_assert(M >= 16 && M <= 1024 && (M & (M - 1)) == 0, "M = n_fft/2 in [16,1024]");
  static constexpr int L = M / 16;             // threads per frame
  static constexpr int FW = 64 / L;            // frames per wave
  static constexpr int REM = M / 16;
  static constexpr int R2 = REM >= 16 ? 16 : REM;  // second radix (1 = no pass)
  static constexpr int R3 = REM / R2;              // third radix (1 = no pass)
  static constexpr int SLOTS = M + M / 16;     // padded complex slots per frame
};
constexpr int WAVE_LDS_SLOTS = 1088;  // 64*16*(17/16) complex = 8704 B, same for every M

__device__ __forceinline__ int phys(int i) { return i + (i >> 4); }

// One Stockham pass on the thread's 16 points: butterflies b use a[b + r*NB].
// Writes results to LDS in autosort order.
template <int R, int NS, int L>
__device__ __forceinline__ void pass_compute_store(float2 (&a)[16], float2* __restrict__ buf, int t,
                                                   const float2* __restrict__ tw /* [NB][R], r=0 unused */) {
  constexpr int NB = 16 / R;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = a[b + r * NB];
    if constexpr (NS > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[b * R + r]);
    }
    Dft<R>::run(v);
    const int j = t + b * L;
    const int o0 = (j / NS) * (NS * R) + (j % NS);
#pragma unroll
    for (int r = 0; r < R; ++r) buf[phys(o0 + r * NS)] = v[r];
  }
}

template <int L>
__device__ __forceinline__ void load_points(float2 (&a)[16], const float2* __restrict__ buf, int t) {
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = buf[phys(t + L * q)];
}

struct StftArgs {
  const float* x;          // (rows, T)
  const float* window;     // (n_fft)
  const float2* tw;        // (n_fft): (cos, -sin)(2 pi k / n_fft)
  float2* out;             // (rows, n_out, M+1) or null
  float* mel;              // (rows, n_out, n_mels) or null
  const int* unit_k0;      // (n_units)
  const float* unit_w;     // (n_units, 16)
  const int* mel_ubeg;     // (n_mels + 1)
  int64_t T;
  int64_t rows;
  int64_t n_out;           // frames written per row
  int frame_lo;            // first frame computed (2 when match_stride drops edges)
  int hop;
  int pad;                 // outer left pad (match_stride)
  int64_t T2;              // T + 2*pad + right_pad
  int pad_mode;
  int chunk;               // frames per wave chunk (multiple of FW)
  int chunks_per_row;
  int n_units;
  int n_mels;
  int vec2;                // 1: float2 input loads are 8-byte aligned
};

// sample fetch with centre reflect padding (torch.stft center=True) applied on
// top of the outer padding (F.pad(audio, (pad, pad+right_pad), mode)).
__device__ __forceinline__ float fetch_padded(const float* __restrict__ xr, int64_t s, const StftArgs& A) {
  // s: index into the outer-padded signal of length T2, may be out of range
  int64_t u = s;
  if (u < 0) u = -u;
  if (u >= A.T2) u = 2 * (A.T2 - 1) - u;
  if (u < 0) u = 0;
  int64_t v = at::pad_index(u - A.pad, A.T, A.pad_mode);
  return v < 0 ? 0.0f : xr[v];
}

template <int M, bool WRITE_STFT, bool MEL>
__global__ __launch_bounds__(256) void stft_mel_kernel(const StftArgs A) {
  using P = Plan<M>;
  constexpr int L = P::L, FW = P::FW, N = 2 * M;
  __shared__ float2 lds[4 * WAVE_LDS_SLOTS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t row = wid / A.chunks_per_row;
  if (row >= A.rows) return;  // whole wave exits together
  const int chunk_id = (int)(wid % A.chunks_per_row);
  const int fs = lane / L;  // frame slot inside the wave
  const int t = lane % L;   // thread inside the frame

  float2* wbuf = lds + wave * WAVE_LDS_SLOTS;
  float2* fbuf = wbuf + fs * P::SLOTS;

  // ---- per-thread constants: window, pass twiddles, split twiddles
  float2 win[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int n = t + L * q;
    win[q] = make_float2(A.window[2 * n], A.window[2 * n + 1]);
  }
  constexpr int NB2 = 16 / P::R2;
  constexpr int NB3 = 16 / P::R3;
  float2 tw2[P::R2 > 1 ? NB2 * P::R2 : 1];
  float2 tw3[P::R3 > 1 ? NB3 * P::R3 : 1];
  if constexpr (P::R2 > 1) {
    constexpr int NS = 16;
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      const int j = t + b * L;
#pragma unroll
      for (int r = 1; r < P::R2; ++r) tw2[b * P::R2 + r] = A.tw[r * (j % NS) * (N / (NS * P::R2))];
    }
  }
  if constexpr (P::R3 > 1) {
    constexpr int NS = 16 * P::R2;
#pragma unroll
    for (int b = 0; b < NB3; ++b) {
      const int j = t + b * L;
#pragma unroll
      for (int r = 1; r < P::R3; ++r) tw3[b * P::R3 + r] = A.tw[r * (j % NS) * (N / (NS * P::R3))];
    }
  }
  float2 twp[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) twp[q] = A.tw[t + L * q];  // (cos, -sin)(2 pi k/N), k = t + L q

  const float* __restrict__ xr = A.x + row * A.T;
  const int f_begin = chunk_id * A.chunk;                       // relative to frame_lo
  const int f_end = min((int64_t)f_begin + A.chunk, A.n_out);

  for (int f0 = f_begin; f0 < f_end; f0 += FW) {
    const int fo = f0 + fs;  // output frame index of this thread's frame
    const bool live = fo < f_end;
    const int64_t frame = (int64_t)fo + A.frame_lo;
    // first sample of the frame in outer-padded coordinates
    const int64_t s0 = frame * A.hop - M;  // M = n_fft/2 centre pad
    float2 a[16];
    if (live) {
      const bool interior = (A.pad == 0) && (s0 >= 0) && (s0 + N <= A.T);
      if (interior) {
        if (A.vec2) {
          const float2* __restrict__ p = reinterpret_cast<const float2*>(xr + s0);
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = p[t + L * q];
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int n = t + L * q;
            a[q] = make_float2(xr[s0 + 2 * n], xr[s0 + 2 * n + 1]);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = t + L * q;
          a[q] = make_float2(fetch_padded(xr, s0 + 2 * n, A), fetch_padded(xr, s0 + 2 * n + 1, A));
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = make_float2(a[q].x * win[q].x, a[q].y * win[q].y);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = make_float2(0.f, 0.f);
    }

    // ---- complex FFT of length M (Stockham, radix 16 / R2 / R3)
    pass_compute_store<16, 1, L>(a, fbuf, t, nullptr);
    wave_sync();
    if constexpr (P::R2 > 1) {
      load_points<L>(a, fbuf, t);
      wave_sync();
      pass_compute_store<P::R2, 16, L>(a, fbuf, t, tw2);
      wave_sync();
    }
    if constexpr (P::R3 > 1) {
      load_points<L>(a, fbuf, t);
      wave_sync();
      pass_compute_store<P::R3, 16 * P::R2, L>(a, fbuf, t, tw3);
      wave_sync();
    }

    // ---- split step: X[k], X[M-k] from Z[k], Z[M-k]; k = t + L q, q < 8
    float2 zk[8], zm[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = t + L * q;
      zk[q] = fbuf[phys(k)];
      zm[q] = fbuf[phys((M - k) & (M - 1))];
    }
    float2 zh = make_float2(0.f, 0.f);
    if (t == 0) zh = fbuf[phys(M / 2)];
    wave_sync();  // all reads of Z done before the slab is reused for |X|

    float* magbuf = reinterpret_cast<float*>(wbuf) + fs * (M + 1);
    float2* __restrict__ orow = nullptr;
    if constexpr (WRITE_STFT) orow = A.out + ((int64_t)row * A.n_out + fo) * (M + 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = t + L * q;
      float2 xa, xb;
      int kb;
      if (q == 0 && t == 0) {  // k == 0: DC and Nyquist
        xa = make_float2(zk[0].x + zk[0].y, 0.f);
        xb = make_float2(zk[0].x - zk[0].y, 0.f);
        kb = M;
      } else {
        const float sr = zk[q].x + zm[q].x, si = zk[q].y - zm[q].y;
        const float dr = zk[q].x - zm[q].x, di = zk[q].y + zm[q].y;
        const float c = twp[q].x, s = -twp[q].y;
        const float pp = fmaf(s, dr, -c * di);
        const float qq = fmaf(s, di, c * dr);
        xa = make_float2(0.5f * (sr - pp), 0.5f * (si - qq));
        xb = make_float2(0.5f * (sr + pp), 0.5f * (-si - qq));
        kb = M - k;
      }
      if constexpr (WRITE_STFT) {
        if (live) {
          orow[k] = xa;
          orow[kb] = xb;
        }
      }
      if constexpr (MEL) {
        magbuf[k] = sqrtf(fmaf(xa.x, xa.x, xa.y * xa.y));
        magbuf[kb] = sqrtf(fmaf(xb.x, xb.x, xb.y * xb.y));
      }
    }
    if (t == 0) {  // k == M/2: X = conj(Z[M/2])
      const float2 xh = make_float2(zh.x, -zh.y);
      if constexpr (WRITE_STFT) {
        if (live) orow[M / 2] = xh;
      }
      if constexpr (MEL) magbuf[M / 2] = sqrtf(fmaf(xh.x, xh.x, xh.y * xh.y));
    }

    if constexpr (MEL) {
      wave_sync();
      float* part = reinterpret_cast<float*>(wbuf) + FW * (M + 1);
      for (int fsl = 0; fsl < FW; ++fsl) {
        const float* mg = reinterpret_cast<float*>(wbuf) + fsl * (M + 1);
        for (int u = lane; u < A.n_units; u += 64) {
          const int k0 = A.unit_k0[u];
          const float4* wq = reinterpret_cast<const float4*>(A.unit_w + (int64_t)u * 16);
          float acc = 0.f;
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const float4 w = wq[i4];
            const int kk = k0 + 4 * i4;
            acc = fmaf(w.x, mg[min(kk + 0, M)], acc);
            acc = fmaf(w.y, mg[min(kk + 1, M)], acc);
            acc = fmaf(w.z, mg[min(kk + 2, M)], acc);
            acc = fmaf(w.w, mg[min(kk + 3, M)], acc);
          }
          part[u] = acc;
        }
        wave_sync();
        const int fo2 = f0 + fsl;
        if (fo2 < f_end) {
          float* mrow = A.mel + ((int64_t)row * A.n_out + fo2) * A.n_mels;
          for (int m = lane; m < A.n_mels; m += 64) {
            const int ub = A.mel_ubeg[m], ue = A.mel_ubeg[m + 1];
            float acc = 0.f;
            for (int u = ub; u < ue; ++u) acc += part[u];
            mrow[m] = acc;
          }
        }
        wave_sync();
      }
    }
  }
}

template <int M>
int launch_m(const StftArgs& A, bool write_stft, bool mel, hipStream_t stream) {
  const int64_t waves = A.rows * A.chunks_per_row;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return AT_ERR_INVALID;
  dim3 grid((unsigned)blocks), block(256);
  if (write_stft && mel)
    hipLaunchKernelGGL((stft_mel_kernel<M, true, true>), grid, block, 0, stream, A);
  else if (write_stft)
    hipLaunchKernelGGL((stft_mel_kernel<M, true, false>), grid, block, 0, stream, A);
  else
    hipLaunchKernelGGL((stft_mel_kernel<M, false, true>), grid, block, 0, stream, A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // namespace

extern "C" {

// Host helper: (cos, -sin)(2 pi k / n_fft), k = 0..n_fft-1, computed in double.
int at_stft_twiddles_host(int n_fft, float* out) {
  if (n_fft <= 0 || out == nullptr) return AT_ERR_INVALID;
  const double w = 6.283185307179586476925286766559 / (double)n_fft;
  for (int k = 0; k < n_fft; ++k) {
    out[2 * k] = (float)cos(w * k);
    out[2 * k + 1] = (float)(-sin(w * k));
  }
  return AT_OK;
}

int at_stft_native_supported(int n_fft) {
  return (n_fft >= 32 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0) ? 1 : 0;
}

int at_stft_mel_f32(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles,
                    int n_fft, int hop, int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out,
                    float* stft_out, const int* mel_unit_k0, const float* mel_unit_w, const int* mel_ubeg,
                    int n_units, int n_mels, float* mel_out, void* stream) {
  if (!x || !window || !twiddles || rows < 0 || T <= 0 || hop <= 0 || pad < 0 || right_pad < 0 ||
      frame_lo < 0 || n_frames_out < 0)
    return AT_ERR_INVALID;
  if (!at_stft_native_supported(n_fft)) return AT_ERR_UNSUPPORTED;
  const bool write_stft = stft_out != nullptr;
  const bool mel = mel_out != nullptr;
  if (!write_stft && !mel) return AT_ERR_INVALID;
  if (mel && (!mel_unit_k0 || !mel_unit_w || !mel_ubeg || n_units <= 0 || n_mels <= 0)) return AT_ERR_INVALID;
  const int M = n_fft / 2;
  const int64_t T2 = T + 2 * (int64_t)pad + right_pad;
  if (M >= T2) return AT_ERR_INVALID;  // torch.stft reflect padding needs n_fft/2 < length
  if (pad_mode == at::PAD_REFLECT && (pad >= T || pad + right_pad >= T)) return AT_ERR_INVALID;
  const int64_t n_total = 1 + T2 / hop;
  if (frame_lo + n_frames_out > n_total) return AT_ERR_INVALID;
  if (rows == 0 || n_frames_out == 0) return AT_OK;
  // LDS budget of the mel stage: FW*(M+1) magnitudes + n_units partials in 2176 floats
  const int FW = 64 / (M / 16);
  if (mel && FW * (M + 1) + n_units > 2 * WAVE_LDS_SLOTS) return AT_ERR_UNSUPPORTED;

  StftArgs A;
  A.x = x; A.window = window; A.tw = reinterpret_cast<const float2*>(twiddles);
  A.out = reinterpret_cast<float2*>(stft_out); A.mel = mel_out;
  A.unit_k0 = mel_unit_k0; A.unit_w = mel_unit_w; A.mel_ubeg = mel_ubeg;
  A.T = T; A.rows = rows; A.n_out = n_frames_out; A.frame_lo = frame_lo; A.hop = hop; A.pad = pad;
  A.T2 = T2; A.pad_mode = pad_mode; A.n_units = n_units; A.n_mels = n_mels;
  A.vec2 = ((T % 2) == 0 && (hop % 2) == 0 && (M % 2) == 0 && (reinterpret_cast<uintptr_t>(x) % 8) == 0) ? 1 : 0;
  // frames per wave chunk: long enough to amortise the per-wave constant loads and keep
  // the overlapping input in L1/L2, short enough to give >= ~4096 waves.
  int64_t chunk = 16 * FW;
  const int64_t want_waves = 4096;
  while (chunk > FW && rows * ((n_frames_out + chunk - 1) / chunk) < want_waves) chunk -= FW;
  A.chunk = (int)chunk;
  A.chunks_per_row = (int)((n_frames_out + chunk - 1) / chunk);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (M) {
    case 16: return launch_m<16>(A, write_stft, mel, s);
    case 32: return launch_m<32>(A, write_stft, mel, s);
    case 64: return launch_m<64>(A, write_stft, mel, s);
    case 128: return launch_m<128>(A, write_stft, mel, s);
    case 256: return launch_m<256>(A, write_stft, mel, s);
    case 512: return launch_m<512>(A, write_stft, mel, s);
    case 1024: return launch_m<1024>(A, write_stft, mel, s);
  }
  return AT_ERR_UNSUPPORTED;
}

}  // extern "C"
