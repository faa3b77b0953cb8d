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


// ---- polyphase resampler on the matrix cores ---------------------------------------------------
// The same sparse polyphase sum as resample_kernel, shaped as the GEMM it is:
//     Y[frame, phase] = sum_tap  X[frame, tap] * Wt[tap, phase],   X[f, k] = xp[f old + k]
// with v_mfma_f32_16x16x4_f32 (exact f32, fma chain): a wave accumulates a tile of 16 frames x 16
// phases, 4 taps per instruction, over the union support window of the phase block (192 of 581
// taps for 441 -> 160).  Why MFMA for an HBM-class kernel: the VALU form needs one LDS read of x
// per 4 FMAs and one weight load per 16 -- LDS (40 % of its cycles bank conflicts), the texture
// addresser and the VALU were co-limiting at 22 % of the HBM roofline.  The MFMA form needs one
// LDS read per 16 FMAs (A operand: one f32 per lane) and one weight float4 per 64, the arithmetic
// moves to the otherwise idle matrix pipe at the same peak rate, and the K-slot permutation
// MFMA_KOFF = {0, 16, 8, 24} makes every A read conflict-free for odd `old`
// (lane (i, k) reads xs[i old + off_k + c]: 32 distinct banks per 32-lane group).
struct ResMfmaArgs {
  const float* x;        // (rows, T)
  const float4* W;       // (NPB, NC, 2, 64) float4: tables.resample_mfma_bank
  const int* lo;         // (NPB): first tap of the phase block's window
  float* out;            // (rows, out_len)
  int64_t T, out_len, rows;
  int old_sr, new_sr, width, NPB, NC;
  int frames_per_tile;   // FT, a multiple of 32
  int tiles_per_row;
  int xs_len;
  int n_load;            // ws kernel: loader waves behind the NPB MFMA waves
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void resample_mfma_kernel(const ResMfmaArgs A) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int FT = A.frames_per_tile;
  const int64_t row = blockIdx.x / A.tiles_per_row;
  const int tile = blockIdx.x % A.tiles_per_row;
  const float* __restrict__ xr = A.x + row * A.T;
  const int64_t f0 = (int64_t)tile * FT;
  const int t = threadIdx.x;
  {  // stage the tile: xs[m] = xp[f0 old + m], xp[m] = x[clamp(m - width)]
    const int64_t g_lo = f0 * A.old_sr - A.width;
    const int n4 = (A.xs_len + 3) / 4;
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
      for (int m = t; m < 4 * n4; m += blockDim.x) {
        int64_t g = g_lo + m;
        g = g < 0 ? 0 : (g >= A.T ? A.T - 1 : g);
        xs[m] = xr[g];
      }
    }
  }
  __syncthreads();
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nwaves = blockDim.x >> 6;
  const int j = lane & 15, k = lane >> 4;
  const int koff = (k & 1) * 16 + (k >> 1) * 8;     // {0, 16, 8, 24}
  const int n_fbp = FT / 32;                        // pairs of 16-frame blocks
  float* __restrict__ orow = A.out + row * A.out_len;
  for (int item = wave; item < A.NPB * n_fbp; item += nwaves) {
    const int P = item % A.NPB;
    const int fp = item / A.NPB;
    const float* __restrict__ a0p = xs + (fp * 32 + j) * A.old_sr + A.lo[P] + koff;   // A[i = j][k]: frame j of block 0
    const float* __restrict__ a1p = a0p + 16 * A.old_sr;                               // frame block 1
    const float4* __restrict__ wp = A.W + (int64_t)P * A.NC * 128 + lane;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 w0 = wp[0], w1 = wp[64];
#pragma unroll 1
    for (int c = 0; c < A.NC; ++c) {
      const int cn = c + 1 < A.NC ? c + 1 : c;       // prefetch the next chunk's weights
      const float4 n0 = wp[cn * 128], n1 = wp[cn * 128 + 64];
      __builtin_amdgcn_sched_barrier(0);             // issue them HERE: 16 MFMAs (512 cycles) cover the L2 latency
      const float* __restrict__ p0 = a0p + 32 * c;
      const float* __restrict__ p1 = a1p + 32 * c;
      float a0[8], a1[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) { a0[s] = p0[s]; a1[s] = p1[s]; }
      const float b[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[s], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      w0 = n0; w1 = n1;
      // keep the prefetched weights in registers: without this the compiler re-materialises them by
      // loading chunk c at the top of iteration c (the bank is const __restrict__), i.e. it waits
      // for L2 in front of the first MFMA of every chunk
      asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(w0.z), "+v"(w0.w), "+v"(w1.x), "+v"(w1.y), "+v"(w1.z), "+v"(w1.w));
    }
    // D: lane holds rows 4 k + r (frames), column j (phase)
    const int ph = 16 * P + j;
    if (ph < A.new_sr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t fa = f0 + fp * 32 + 4 * k + r;
        const int64_t oa = fa * A.new_sr + ph, ob = oa + 16 * (int64_t)A.new_sr;
        if (oa < A.out_len) orow[oa] = acc0[r];
        if (ob < A.out_len) orow[ob] = acc1[r];
      }
    }
  }
}


// ---- the same, wave-specialised and persistent ("ws") ----------------------------------------------
// resample_mfma_kernel alternates "stage a tile" and "compute it" per workgroup, and its MFMA pipe
// is busy ~46 % of the launch.  Here the roles are separate waves of a persistent workgroup:
//   * wave p < NPB owns phase block p for the whole launch: its weight window (2 float4 per 32-tap
//     chunk) is loaded ONCE into registers, so its instruction stream is LDS reads + MFMAs only;
//   * one LOADER wave moves the next 16-frame tile HBM -> registers -> the other LDS buffer while the
//     MFMA waves work on the current one (its vmcnt is its own: no in-order coupling with the
//     weight loads, which is what sank the single-role persistent variant);
//   * one __syncthreads() per tile swaps the buffers.
// Tiles are 16 frames (one MFMA row block): two buffers of ~30 KB, two workgroups per CU, and
// 2 x NPB MFMA waves per CU spread evenly over the four SIMDs.
// Where its time goes (round 3, profiles/r03_notes.md): 375 tiles per CU at 3.3 us each against 1.8 us of matrix-pipe
// work.  With 85 registers per wave (48 of them the weights) the compiler single-buffers the A operands --
// ds_read2_b32, s_waitcnt lgkmcnt(0), two MFMAs, next read --, and a workgroup has ONE tile load in flight: issued after
// the barrier, ~28 DMA pieces plus an HBM round trip before the next barrier can fall.  Tried: 2 - 4 loader waves
// (AT_RESAMPLE_LOADERS: 1.19 / 1.41 / 1.40 ms against 1.21), and one workgroup per CU with a 170-register budget
// and all 48 operands of a tile read before the first MFMA (1.48 ms against 1.27-1.32: the second workgroup hides
// more than the deeper pipeline gains).  A ring of three tile buffers does not fit twice into 160 KB; as ONE workgroup
// per CU -- three buffers, two loader waves alternating so that two tile loads are in flight with two tile periods
// each, operands in registers -- it measured 1.45-1.49 ms against 1.28-1.32 (profiles/r03_s72_resample_ws3.txt): the
// load latency is not what holds the kernel either; ten matrix waves per CU keep the pipe ~47 % busy, twenty 57 %.
constexpr int WS_NC = 6;          // 32-tap chunks a wave keeps in registers (441 -> 160 needs 6)

template <int NC>
__global__ __launch_bounds__(1024, 6) void resample_mfma_ws_kernel(const ResMfmaArgs A) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave >= A.NPB;
  const int lw = wave - A.NPB, nl = A.n_load;   // loader waves deal the 1 KB pieces of a tile round-robin
  const int n4 = (A.xs_len + 3) / 4;
  const int n4s = (n4 + 63) / 64 * 64;          // whole 1 KB DMA pieces
  // the two tile buffers are addressed as xs + cur * buf_stride: an array of pointers indexed by a
  // run-time value decays to generic (flat) pointers and the MFMA operand reads became flat loads
  const int buf_stride = 4 * n4s;
  const int64_t n_tiles = A.rows * (int64_t)A.tiles_per_row;
  const int j = lane & 15, k = lane >> 4;
  const int koff = (k & 1) * 16 + (k >> 1) * 8;

  // LDS-DMA (global_load_lds_dwordx4): 1 KB per wave instruction straight into the LDS buffer, no
  // staging registers -- the loader wave keeps a whole tile in flight with a handful of VGPRs
  // (a register-staged tile needs ~30 float4 per lane and pushed the kernel past 5 waves per SIMD).
  auto load_tile = [&](int64_t tile_id, float* __restrict__ dst) __attribute__((always_inline)) {
    const int64_t row = tile_id / A.tiles_per_row;
    const int tile = (int)(tile_id - row * A.tiles_per_row);
    const float* __restrict__ xr = A.x + row * A.T;
    const int64_t g_lo = (int64_t)tile * 16 * A.old_sr - A.width;
    if (g_lo >= 0 && g_lo + 4 * (int64_t)n4 <= A.T) {
      const float* __restrict__ src = xr + g_lo;
      for (int i0 = 64 * lw; i0 < n4; i0 += 64 * nl) {
        const int i = i0 + lane;
        const float* g = src + 4 * (i < n4 ? i : 0);       // lanes past the end re-read element 0 into the slack
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + 4 * i0), 16, 0, 0);
      }
    } else {
      for (int m = lane + 64 * lw; m < 4 * n4; m += 64 * nl) {
        int64_t g = g_lo + m;
        g = g < 0 ? 0 : (g >= A.T ? A.T - 1 : g);
        dst[m] = xr[g];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // LDS-DMA is ordered by the issuing wave's vmcnt only
  };

  // weights of this wave's phase block: registers for the whole launch
  float4 wr[2 * NC];
  if (!loader) {
    const float4* __restrict__ wp = A.W + (int64_t)wave * NC * 128 + lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) { wr[2 * c] = wp[c * 128]; wr[2 * c + 1] = wp[c * 128 + 64]; }
  }
  int cur = 0;
  int64_t tile_id = blockIdx.x;
  if (loader && tile_id < n_tiles) load_tile(tile_id, xs);
  __syncthreads();
  const int lo_p = loader ? 0 : A.lo[wave];
  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    if (loader) {
      const int64_t next = tile_id + gridDim.x;
      if (next < n_tiles) load_tile(next, xs + (cur ^ 1) * buf_stride);
      __syncthreads();
    } else {
      const float* __restrict__ ap = xs + cur * buf_stride + j * A.old_sr + lo_p + koff;   // A[i = j][k]
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // even / odd MFMAs: two dependency chains
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float a[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) a[s2] = ap[32 * c + s2];
        const float b[8] = {wr[2 * c].x, wr[2 * c].y, wr[2 * c].z, wr[2 * c].w,
                            wr[2 * c + 1].x, wr[2 * c + 1].y, wr[2 * c + 1].z, wr[2 * c + 1].w};
#pragma unroll
        for (int s2 = 0; s2 < 8; s2 += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2], b[s2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2 + 1], b[s2 + 1], acc1, 0, 0, 0);
        }
      }
      // the tile buffer is free once the operands are in registers: rendezvous FIRST, store the
      // results afterwards (the stores of this tile then overlap the next tile's operand reads of
      // the other waves instead of holding everybody at the barrier)
      __syncthreads();
      const int64_t row = tile_id / A.tiles_per_row;
      const int64_t f0 = (tile_id - row * A.tiles_per_row) * 16;
      float* __restrict__ orow = A.out + row * A.out_len;
      const int ph = 16 * wave + j;
      if (ph < A.new_sr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t o = (f0 + 4 * k + r) * A.new_sr + ph;
          if (o < A.out_len) orow[o] = acc0[r] + acc1[r];
        }
      }
    }
    cur ^= 1;
  }
}


// ---- per-item filter design on the device -------------------------------------------------------------------------
// low_pass / high_pass (dsp.py:177-179 -> julius.LowPassFilter: Hann-windowed sinc of half size int(zeros / c / 2),
// unit DC gain) and equalizer (effects.py:399-432 -> julius.SplitBands bank, weighted band sum) designed ONE launch per
// call, straight into the zero-padded (B, L_padded) tap table the FIR kernels read.  The torch formulation they replace
// is ~25 (sinc) / ~8 (equalizer) whole-table elementwise launches of 3-5 us each per call -- with a (B, bands, L)
// intermediate for the equalizer -- i.e. as long on the GPU as the filtering itself at cfg4 (1.03 ms for a 0.51 ms
// kernel).  Arithmetic: float32, the operation order of the torch formulation (kernels.sinc_taps_batched, fx.equalizer_taps).
constexpr int DESIGN_THREADS = 256;

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  const float tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return tot;
}

// cutoffs (B) normalised (cycles / sample); row b: taps of length 2 H + 1 centred at column H (its own filter has
// half size h_b <= H, zero outside), columns [2 H + 1, Lp) zero.
__global__ __launch_bounds__(DESIGN_THREADS) void sinc_taps_kernel(const float* __restrict__ cutoffs, float zeros, int H, int Lp,
                                                                   float* __restrict__ taps) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float c = cutoffs[b];
  const bool pos = c > 0.f;
  const int half = pos ? (int)((zeros / c) / 2.0f) : 0;
  const float hf = (float)half;
  const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
  const float den = fmaxf(2.0f * hf, 1.0f);
  float* __restrict__ row = taps + (int64_t)b * Lp;
  const int L = 2 * H + 1;
  float part = 0.f;
  for (int i = threadIdx.x; i < L; i += DESIGN_THREADS) {
    const int n = i - H;
    const float nf = (float)n;
    const bool inside = (n < 0 ? -n : n) <= half;
    float win = 0.5f - 0.5f * cosf(two_pi * (nf + hf) / den);
    if (half == 0) win = 1.0f;
    const float arg = ((2.0f * c) * pi) * nf;
    const float sinc = arg == 0.f ? 1.0f : sinf(arg) / arg;
    const float h = inside ? ((2.0f * c) * win) * sinc : 0.f;
    row[i] = h;
    part += h;
  }
  const float tot = block_sum_256(part, sh);
  for (int i = threadIdx.x; i < Lp; i += DESIGN_THREADS) {
    float h = i < L ? row[i] / tot : 0.f;                    // (a thread re-reads only what it wrote)
    if (!pos) h = 0.f;
    row[i] = h;
  }
}

// weights (B, n_bands); bank (n_bands - 1, L) low-pass bank of the band split; taps[b] = sum_k (w[b,k] - w[b,k+1]) bank[k]
// + w[b, last] delta(half), zero-padded to Lp.
__global__ __launch_bounds__(DESIGN_THREADS) void eq_taps_kernel(const float* __restrict__ weights, const float* __restrict__ bank,
                                                                 int n_bands, int L, int half, int Lp, float* __restrict__ taps) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * DESIGN_THREADS + threadIdx.x;
  if (i >= Lp) return;
  const float* __restrict__ w = weights + (int64_t)b * n_bands;
  float acc = 0.f;
  if (i < L) {
    for (int k = 0; k + 1 < n_bands; ++k) acc += (w[k] - w[k + 1]) * bank[(int64_t)k * L + i];
    if (i == half) acc += w[n_bands - 1];
  }
  taps[(int64_t)b * Lp + i] = acc;
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


int at_resample_mfma_supported(int old_sr, int new_sr) {
  // conflict-free A reads need an odd (gcd-reduced) source rate; even ones keep the VALU kernel
  return (old_sr > 0 && new_sr > 0 && (old_sr & 1)) ? 1 : 0;
}

// MFMA form of at_resample_f32 (same result up to the order of the float32 sums).
//   W, lo, NPB, NC: tables.resample_mfma_bank(old, new);  max_lo = max(lo).
int at_resample_mfma_f32(const float* x, int64_t rows, int64_t T, const float* W, const int* lo, int old_sr, int new_sr,
                         int width, int NPB, int NC, int max_lo, float* out, int64_t out_len, void* stream) {
  if (rows == 0) return AT_OK;
  if (!x || !W || !lo || !out || rows < 0 || T <= 0 || old_sr <= 0 || new_sr <= 0 || width <= 0 || NPB <= 0 || NC <= 0 ||
      max_lo < 0 || out_len < 0 || 16 * NPB < new_sr)
    return AT_ERR_INVALID;
  if (!at_resample_mfma_supported(old_sr, new_sr)) return AT_ERR_UNSUPPORTED;
  if (out_len == 0) return AT_OK;
  ResMfmaArgs A;
  A.x = x; A.W = reinterpret_cast<const float4*>(W); A.lo = lo; A.out = out; A.T = T; A.out_len = out_len; A.rows = rows;
  A.old_sr = old_sr; A.new_sr = new_sr; A.width = width; A.NPB = NPB; A.NC = NC; A.n_load = 1;
  const int64_t frames = (out_len + new_sr - 1) / new_sr;
  static const int use_ws = at::env_int_once("AT_RESAMPLE_WS", 1);
  if (use_ws && NPB <= 15 && NC <= WS_NC) {
    // wave-specialised persistent form: 16-frame tiles, two LDS buffers
    A.frames_per_tile = 16;
    A.tiles_per_row = (int)((frames + 15) / 16);
    A.xs_len = 16 * old_sr + max_lo + 32 * NC + 32;
    const size_t lds1 = (size_t)(((A.xs_len + 3) / 4 + 63) / 64 * 64) * 16;   // whole 1 KB LDS-DMA pieces
    if (2 * lds1 <= 80 * 1024) {
      // one loader wave issues a tile's ~28 LDS-DMA pieces back to back (60-185 cycles each, MI355X_MICROARCH.md) and
      // then waits for the last: longer than the MFMA waves need for a tile; AT_RESAMPLE_LOADERS deals them over more waves
      static const int n_load_env = at::env_int_once("AT_RESAMPLE_LOADERS", 1);
      A.n_load = n_load_env < 1 ? 1 : (n_load_env > 16 - NPB ? 16 - NPB : n_load_env);
      const int threads = (NPB + A.n_load) * 64;
      const int64_t tiles = rows * A.tiles_per_row;
      int per_cu = (int)((160 * 1024) / (2 * lds1));
      if (per_cu > 2048 / threads) per_cu = 2048 / threads;
      if (per_cu > 2) per_cu = 2;
      if (per_cu < 1) per_cu = 1;
      int64_t blocks = (int64_t)at::device_cu_count() * per_cu;
      if (blocks > tiles) blocks = tiles;
      hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define AT_WS_LAUNCH(NCV)                                                                                  \
  case NCV: {                                                                                              \
    int e = at::allow_big_lds(reinterpret_cast<const void*>(resample_mfma_ws_kernel<NCV>));                \
    if (e != AT_OK) return e;                                                                              \
    hipLaunchKernelGGL(resample_mfma_ws_kernel<NCV>, dim3((unsigned)blocks), dim3(threads), 2 * lds1, st, A); \
    break;                                                                                                 \
  }
      switch (NC) {
        AT_WS_LAUNCH(1) AT_WS_LAUNCH(2) AT_WS_LAUNCH(3) AT_WS_LAUNCH(4) AT_WS_LAUNCH(5) AT_WS_LAUNCH(6)
        default: return AT_ERR_UNSUPPORTED;
      }
#undef AT_WS_LAUNCH
      AT_LAUNCH_CHECK();
      return AT_OK;
    }
  }
  // tile: ~14.5k input samples of LDS (58 KB -> 2 workgroups per CU), a multiple of 32 frames
  int FT = 14500 / old_sr / 32 * 32;
  if (FT < 32) FT = 32;
  if (FT > 1024) FT = 1024;
  if (FT > (frames + 31) / 32 * 32) FT = (int)((frames + 31) / 32 * 32);
  A.frames_per_tile = FT;
  A.tiles_per_row = (int)((frames + FT - 1) / FT);
  A.xs_len = FT * old_sr + max_lo + 32 * NC + 32;
  const size_t lds = (size_t)((A.xs_len + 3) / 4 * 4) * 4;
  if (lds > 160 * 1024) return AT_ERR_UNSUPPORTED;
  {
    int e = at::allow_big_lds(reinterpret_cast<const void*>(resample_mfma_kernel));
    if (e != AT_OK) return e;
  }
  const int64_t blocks = rows * A.tiles_per_row;
  if (blocks > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;
  int64_t items = (int64_t)NPB * (FT / 32);
  int threads = (int)(items < 16 ? items : 16) * 64;
  // at least 4 waves so that staging a tile is not a one-wave job
  if (threads < 256) threads = 256;
  hipLaunchKernelGGL(resample_mfma_kernel, dim3((unsigned)blocks), dim3(threads), lds, reinterpret_cast<hipStream_t>(stream), A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}


// Device-side design of the per-item windowed-sinc low-pass taps of low_pass / high_pass (dsp.py:177-179, 209-211 ->
// julius.LowPassFilter).  cutoffs (B) f32 normalised cutoffs in (0, 0.5] (0 = the all-zero filter; range checks are the
// caller's), zeros = julius' `zeros`, H = max_b int(zeros / c_b / 2) (the common half size, computed by the caller from
// its host copy of the cutoffs), taps (B, L_padded) with L_padded >= 2 H + 1: row b = the filter of item b centred at
// column H, zero elsewhere -- the table at_fir_fft_f32 / at_fir_per_item_f32 take with half = H.
int at_sinc_taps_f32(const float* cutoffs, int64_t B, float zeros, int H, int L_padded, float* taps, void* stream) {
  if (B == 0) return AT_OK;
  if (!cutoffs || !taps || B < 0 || B > 0x7fffffffLL || H < 0 || L_padded < 2 * H + 1 || !(zeros > 0.f)) return AT_ERR_INVALID;
  hipLaunchKernelGGL(sinc_taps_kernel, dim3((unsigned)B), dim3(DESIGN_THREADS), 0, reinterpret_cast<hipStream_t>(stream), cutoffs,
                     zeros, H, L_padded, taps);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// Composite equalizer FIR per item (effects.py:399-403, 429-432: julius.SplitBands + weighted band sum collapsed to one
// filter): weights (B, n_bands) linear band gains, bank (n_bands - 1, L) the low-pass bank of the band split (all rows
// of the common odd length L, centre `half`), taps (B, L_padded >= L).
int at_eq_taps_f32(const float* weights, const float* bank, int64_t B, int n_bands, int L, int half, int L_padded, float* taps,
                   void* stream) {
  if (B == 0) return AT_OK;
  if (!weights || !bank || !taps || B < 0 || B > 65535 || n_bands < 2 || L <= 0 || half < 0 || half >= L || L_padded < L)
    return AT_ERR_INVALID;
  const dim3 grid((unsigned)((L_padded + DESIGN_THREADS - 1) / DESIGN_THREADS), (unsigned)B);
  hipLaunchKernelGGL(eq_taps_kernel, grid, dim3(DESIGN_THREADS), 0, reinterpret_cast<hipStream_t>(stream), weights, bank, n_bands,
                     L, half, L_padded, taps);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // extern "C"
