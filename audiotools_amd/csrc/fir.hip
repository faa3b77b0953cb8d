// Time-domain FIR kernels for gfx950: per-item windowed-sinc filtering and polyphase resampling.
//
// at_fir_per_item_f32 replaces the reference's per-item Python loops
//   audiotools/core/dsp.py:177-179, 209-211   julius.LowPassFilter / HighPassFilter per batch item
//   audiotools/core/effects.py:399-403,429-432 julius.SplitBands + weighted band sum (equalizer),
//                                              collapsed to ONE composite FIR per item
// at_resample_f32 replaces
//   audiotools/core/audio_signal.py:732        julius.resample_frac (replicate pad + strided conv1d)
//
// FIR (direct form, used for short filters; long ones go to the overlap-save kernel in
// firfft.hip): one workgroup = 4096 consecutive outputs of one (item, channel) row.  The input
// window (with replicate padding applied while staging) and the item's taps sit in LDS; every
// thread owns 16 consecutive outputs and slides a 24-sample register window over the taps,
// 128 FMAs per 2 ds_read_b128 of x -- bound by the FP32 vector rate, not by LDS or HBM.
// Resampler: see the comment above resample_kernel.
#include "at_common.h"

namespace {

constexpr int FIR_R = 16;        // consecutive outputs per thread
constexpr int FIR_TILE = 256 * FIR_R;   // outputs per workgroup
constexpr int FIR_CHUNK = 512;   // taps staged per pass
constexpr int FIR_XS = (FIR_TILE + FIR_CHUNK + 32) / 16 * 20;
// LDS window is stored in rows of 16 samples padded to 20 floats (80 B): the per-lane stride of the
// ds_read_b128 window reads becomes 80 B instead of 64 B -> conflict-free
__device__ __forceinline__ int fir_pad(int n) { return n + 4 * (n >> 4); }

__global__ __launch_bounds__(256) void fir_per_item_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                           float* __restrict__ out, int64_t T, int C, int taps_rows,
                                                           int Lp /* padded to a multiple of 8 */, int half, int highpass,
                                                           int tiles_per_row) {
  __shared__ __attribute__((aligned(16))) float xs[FIR_XS];
  __shared__ __attribute__((aligned(16))) float hs[FIR_CHUNK];
  const int64_t row = blockIdx.x / tiles_per_row;
  const int tile = blockIdx.x % tiles_per_row;
  const int64_t item = row / C;
  const float* __restrict__ xr = x + row * T;
  const float* __restrict__ h = taps + (taps_rows == 1 ? 0 : item) * (int64_t)Lp;
  const int64_t n0 = (int64_t)tile * FIR_TILE;
  const int t = threadIdx.x;

  float acc[FIR_R];
#pragma unroll
  for (int i = 0; i < FIR_R; ++i) acc[i] = 0.f;

  for (int j0 = 0; j0 < Lp; j0 += FIR_CHUNK) {
    const int nj = min(FIR_CHUNK, Lp - j0);  // multiple of 8
    __syncthreads();
    // stage x[n0 + j0 - half + m], m in [0, FIR_TILE + nj + 8), replicate padding at both ends
#pragma unroll 4
    for (int m = t; m < FIR_TILE + nj + 16; m += 256) {
      int64_t g = n0 + j0 - half + m;
      g = g < 0 ? 0 : (g >= T ? T - 1 : g);
      xs[fir_pad(m)] = xr[g];
    }
    for (int m = t; m < nj; m += 256) hs[m] = h[j0 + m];
    __syncthreads();

    // register window: xw[i] = xs[R t + j + i], i < R + 8; slides by 8 taps per step
    float xw[FIR_R + 8];
#pragma unroll
    for (int i4 = 0; i4 < FIR_R / 4; ++i4) {
      const float4 a = *reinterpret_cast<const float4*>(xs + fir_pad(FIR_R * t + 4 * i4));
      xw[8 + 4 * i4 + 0] = a.x; xw[8 + 4 * i4 + 1] = a.y; xw[8 + 4 * i4 + 2] = a.z; xw[8 + 4 * i4 + 3] = a.w;
    }
    for (int j = 0; j < nj; j += 8) {
#pragma unroll
      for (int i = 0; i < FIR_R; ++i) xw[i] = xw[i + 8];
      const float4 a = *reinterpret_cast<const float4*>(xs + fir_pad(FIR_R * t + j + FIR_R));
      const float4 b = *reinterpret_cast<const float4*>(xs + fir_pad(FIR_R * t + j + FIR_R + 4));
      xw[FIR_R + 0] = a.x; xw[FIR_R + 1] = a.y; xw[FIR_R + 2] = a.z; xw[FIR_R + 3] = a.w;
      xw[FIR_R + 4] = b.x; xw[FIR_R + 5] = b.y; xw[FIR_R + 6] = b.z; xw[FIR_R + 7] = b.w;
      const float4 h0 = *reinterpret_cast<const float4*>(hs + j);
      const float4 h1 = *reinterpret_cast<const float4*>(hs + j + 4);
      const float hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < FIR_R; ++i) acc[i] = fmaf(hh[u], xw[i + u], acc[i]);
    }
  }
  float* __restrict__ orow = out + row * T;
#pragma unroll
  for (int i = 0; i < FIR_R; ++i) {
    const int64_t n = n0 + FIR_R * t + i;
    if (n < T) orow[n] = highpass ? xr[n] - acc[i] : acc[i];
  }
}

// ---- polyphase resampler ------------------------------------------------------------------
// y[f*new + i] = sum_m wg[G][m][p] * xp[f*old + base[G] + m],   i = 4 G + p
// The julius bank is banded: phase i only has taps in a window of ~2*zeros*old/sr that slides with
// i.  Phases are grouped by 4; a group stores its taps densely over the union window (LG taps,
// zero filled) as float4 = 4 phases per tap.  A thread owns one group x RS_FB consecutive frames:
// 16 FMAs per 4 LDS reads of x + one float4 of weights (L2 resident, ~100 KB per ratio).
// The bank is tap-major (LG, NG): the lanes of a wave hold adjacent groups, so a weight load is a
// few contiguous cache lines per wave instruction (group-major made every lane touch its own
// line: 64 lines per instruction, the texture-address unit was the bottleneck).  The workgroup
// size is the work-item count of a tile rounded up to whole waves (one round, no idle tail).
struct ResampleArgs {
  const float* x;        // (rows, T)
  const float4* wg;      // (LG, NG): 4 phases per tap, tap-major
  const int* base;       // (NG): first dense tap index of the group
  float* out;            // (rows, out_len)
  int64_t T, out_len, rows;
  int old_sr, new_sr, width, NG, LG;
  int frames_per_tile;   // FT (multiple of RS_FB)
  int tiles_per_row;
  int xs_len;
};

typedef float v4f __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };  // dword-aligned quad

constexpr int RS_FB = 4;  // frames per thread
constexpr int RS_TB = 4;  // taps per pipelined block (LG is padded to a multiple)

__global__ __launch_bounds__(1024) void resample_kernel(const ResampleArgs A) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int FT = A.frames_per_tile;
  const int64_t row = blockIdx.x / A.tiles_per_row;
  const int tile = blockIdx.x % A.tiles_per_row;
  const float* __restrict__ xr = A.x + row * A.T;
  const int64_t f0 = (int64_t)tile * FT;
  const int t = threadIdx.x;
  // padded input: xp[m] = x[clamp(m - width)], frame f reads xp[f*old + k].
  // Interior tiles: unaligned 16-byte loads, several in flight per thread (a one-load-per-
  // iteration loop exposes the full HBM latency ~40 times per tile and was 80 % of the kernel).
  {
    const int64_t g_lo = f0 * A.old_sr - A.width;
    const int n4 = (A.xs_len + 3) / 4;  // the LDS allocation is rounded up to whole float4
    if (g_lo >= 0 && g_lo + 4 * (int64_t)n4 <= A.T) {
      const f4u* __restrict__ src = reinterpret_cast<const f4u*>(xr + g_lo);
      float4* __restrict__ dst = reinterpret_cast<float4*>(xs);
#pragma unroll 4
      for (int i = t; i < n4; i += blockDim.x) {
        const f4u v = src[i];
        dst[i] = make_float4(v.x, v.y, v.z, v.w);
      }
    } else {
#pragma unroll 4
      for (int m = t; m < A.xs_len; m += blockDim.x) {
        int64_t g = g_lo + m;
        g = g < 0 ? 0 : (g >= A.T ? A.T - 1 : g);
        xs[m] = xr[g];
      }
    }
  }
  __syncthreads();
  const int fgroups = FT / RS_FB;
  float* __restrict__ orow = A.out + row * A.out_len;
  for (int item = t; item < A.NG * fgroups; item += blockDim.x) {
    const int G = item % A.NG;
    const int fg = item / A.NG;
    // acc[r] = 4 phases of frame r as one 4-vector: the FMA below becomes two v_pk_fma_f32 whose
    // A operand is the weight register pair as loaded and whose B operand is x broadcast by
    // op_sel -- no register shuffling (pairing across frames costs a v_mov per FMA pair).
    v4f acc[RS_FB];
#pragma unroll
    for (int r = 0; r < RS_FB; ++r) acc[r] = (v4f)(0.f);
    const float* xb = xs + (fg * RS_FB) * A.old_sr + A.base[G];
    const v4f* __restrict__ w = reinterpret_cast<const v4f*>(A.wg) + G;
    // LG is a multiple of RS_TB (host pads with zero weights)
#pragma unroll 1
    for (int m0 = 0; m0 < A.LG; m0 += RS_TB) {
      v4f wv[RS_TB];
      float xv[RS_TB][RS_FB];
#pragma unroll
      for (int u = 0; u < RS_TB; ++u) {
        wv[u] = w[(int64_t)(m0 + u) * A.NG];
#pragma unroll
        for (int r = 0; r < RS_FB; ++r) xv[u][r] = xb[r * A.old_sr + m0 + u];
      }
#pragma unroll
      for (int u = 0; u < RS_TB; ++u)
#pragma unroll
        for (int r = 0; r < RS_FB; ++r) acc[r] = __builtin_elementwise_fma(wv[u], (v4f)(xv[u][r]), acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RS_FB; ++r) {
      const int64_t f = f0 + fg * RS_FB + r;
      const int64_t o = f * A.new_sr + 4 * G;      // the thread's 4 phases are 4 consecutive outputs
      if (4 * G + 3 < A.new_sr && o + 3 < A.out_len) {
        f4u v; v.x = acc[r].x; v.y = acc[r].y; v.z = acc[r].z; v.w = acc[r].w;
        *reinterpret_cast<f4u*>(orow + o) = v;
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (4 * G + p < A.new_sr && o + p < A.out_len) orow[o + p] = acc[r][p];
      }
    }
  }
}

}  // namespace

extern "C" {

// x (B,C,T); taps (taps_rows, L_padded) with taps_rows == 1 or B, each row a CENTRED odd-length
// FIR zero-padded to L_padded (a multiple of 8, centre tap at index `half`); replicate padding.
// out = FIR(x) or, with highpass != 0, x - FIR(x).
int at_fir_per_item_f32(const float* x, int64_t B, int64_t C, int64_t T, const float* taps, int taps_rows,
                        int L_padded, int half, int highpass, float* out, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !taps || !out || B < 0 || C <= 0 || T <= 0 || L_padded <= 0 || (L_padded % 8) != 0 || half < 0 ||
      half >= L_padded || (taps_rows != 1 && taps_rows != B))
    return AT_ERR_INVALID;
  if (B == 0) return AT_OK;
  const int64_t tiles = (T + FIR_TILE - 1) / FIR_TILE;
  const int64_t blocks = B * C * tiles;
  if (blocks > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fir_per_item_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     taps, out, T, (int)C, taps_rows, L_padded, half, highpass, (int)tiles);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// x (rows,T) -> out (rows,out_len), out_len = floor(new_sr*T/old_sr) for the REDUCED ratio.
// wg (LG, NG, 4) f32 tap-major, base (NG) i32: the bank grouped 4 phases per tap (tables.resample_grouped_bank).
int at_resample_f32(const float* x, int64_t rows, int64_t T, const float* wg, const int* base, int old_sr, int new_sr,
                    int width, int NG, int LG, float* out, int64_t out_len, void* stream) {
  if (rows == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !wg || !base || !out || rows < 0 || T <= 0 || old_sr <= 0 || new_sr <= 0 || width <= 0 || NG <= 0 ||
      LG <= 0 || (LG % RS_TB) != 0 || out_len < 0 || 4 * NG < new_sr)
    return AT_ERR_INVALID;
  if (rows == 0 || out_len == 0) return AT_OK;
  ResampleArgs A;
  A.x = x; A.wg = reinterpret_cast<const float4*>(wg); A.base = base; A.out = out; A.T = T; A.out_len = out_len;
  A.rows = rows; A.old_sr = old_sr; A.new_sr = new_sr; A.width = width; A.NG = NG; A.LG = LG;
  const int64_t frames = (out_len + new_sr - 1) / new_sr;
  // tile: ~14k input samples of LDS (56 KB -> 2 workgroups per CU), a multiple of RS_FB frames
  int FT = 14000 / old_sr / RS_FB * RS_FB;
  if (FT < RS_FB) FT = RS_FB;
  if (FT > 64) FT = 64;
  A.frames_per_tile = FT;
  A.tiles_per_row = (int)((frames + FT - 1) / FT);
  // last frame of the tile reads up to (FT-1)*old + max(base)+LG-1 <= (FT-1)*old + 2*width + old
  A.xs_len = FT * old_sr + 2 * width + LG;
  const size_t lds = (size_t)((A.xs_len + 3) / 4 * 4) * 4;
  if (lds > 160 * 1024) return AT_ERR_UNSUPPORTED;
  {
    int e = at::allow_big_lds(reinterpret_cast<const void*>(resample_kernel));
    if (e != AT_OK) return e;
  }
  const int64_t blocks = rows * A.tiles_per_row;
  if (blocks > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;
  int threads = (NG * (FT / RS_FB) + 63) / 64 * 64;   // one round of work items, whole waves
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(threads), lds, reinterpret_cast<hipStream_t>(stream), A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // extern "C"
