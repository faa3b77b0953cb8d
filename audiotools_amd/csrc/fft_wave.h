// Wave-level FFT building blocks shared by stft.hip and istft.hip (gfx950).
// Every thread owns 16 complex points; L = M/16 threads form one M-point transform, a wave64
// runs 64/L transforms at once.  Stockham autosort passes of radix 16 / R2 / R3 in registers,
// exchanges through a per-wave LDS slab, bank-swizzled by phys<L>().
#pragma once
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

// LDS slot of logical point i of an M = 16 L point frame.
//  L <= 16 (several frames per wave): i + i/16 -- with the odd frame stride M + M/16 the stride-16
//           stores of pass 1 and the contiguous reads are conflict-free.
//  L >= 32: XOR swizzle of the low 4 bits with the 16-row index.  A permutation inside every
//           16-row, so 32 contiguous lanes of a ds_read_b64 still cover all 64 banks (the +i/16
//           padding made those reads wrap onto banks 0,1: 2-way conflict on EVERY exchange read,
//           23 % of all LDS cycles in the r01 profile), and the 16 lanes of a pass-1
//           ds_write_b64 (index 16 t + r) land on 16 distinct bank pairs.
#ifndef AT_FFT_SWIZZLE
#define AT_FFT_SWIZZLE 1
#endif
// PAD256 (L = 64 only): one extra slot per 256 points.  Every exchange instruction of the wave FFT
// touches a single 256-block, so the shift is uniform per instruction (no new conflicts), but the
// stride-256 reads of the last pass are no longer 2048 B apart and the compiler cannot fuse them
// into ds_read2st64_b64 (half the rate of two ds_read_b64, MI355X_MICROARCH.md LDS table).
template <int L, bool PAD256 = false>
__device__ __forceinline__ int phys(int i) {
  if constexpr (L >= 32 && AT_FFT_SWIZZLE) return (i ^ ((i >> 4) & 15)) + (PAD256 ? (i >> 8) : 0);
  else return i + (i >> 4);
}

// |z| with the hardware v_sqrt_f32 (1 ulp): the IEEE sqrtf() expansion costs ~10x more VALU
__device__ __forceinline__ float cabs_fast(float2 z) { return __builtin_amdgcn_sqrtf(fmaf(z.x, z.x, z.y * z.y)); }

// value of lane (i + n) inside the 16-lane DPP row; 0 past the end of the row
template <int n>
__device__ __forceinline__ float dpp_row_shl(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 | n, 0xf, 0xf, true));
}

// One Stockham pass on the thread's 16 points: butterflies b use a[b + r*NB].
// Writes results to LDS in autosort order.
template <int R, int NS, int L, bool PAD256 = false>
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
    for (int r = 0; r < R; ++r) buf[phys<L, PAD256>(o0 + r * NS)] = v[r];
  }
}

// The LAST pass of a transform (NS * R == M) needs no exchange: butterfly j = t + b L writes points
// j + r NS = t + L (b + r NB), i.e. exactly the register slots q = b + r NB of the same thread in the
// "point t + L q" layout that load_points() produces.  Results stay in a[].
template <int R, int NS, int L>
__device__ __forceinline__ void pass_compute_regs(float2 (&a)[16], const float2* __restrict__ tw /* [NB][R], r=0 unused */) {
  static_assert(NS * R == 16 * L, "only the last pass is in place");
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
#pragma unroll
    for (int r = 0; r < R; ++r) a[b + r * NB] = v[r];
  }
}

template <int L, bool PAD256 = false>
__device__ __forceinline__ void load_points(float2 (&a)[16], const float2* __restrict__ buf, int t) {
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = buf[phys<L, PAD256>(t + L * q)];
}


}  // namespace
