// ITU-R BS.1770 integrated loudness for gfx950.
//
// Replaces reference audiotools/core/loudness.py:
//   :102-126  apply_filter_cpu  (cascade of float32 Direct-Form-I biquads, the CPU/IIR branch)
//   :164-174  _unfold           (400 ms blocks, 75 % overlap, zero-padded tail)
//   :176-247  integrated_loudness (block energies, absolute + relative gating)
//
// Kernel 1  kweight_hop_energy: one wave64 per (row, segment of hops).
//   The wave walks its segment in super-blocks of 2048 samples: the 8 KB tile is read
//   from HBM with 16-byte-per-lane coalesced loads, transposed through a padded LDS slab,
//   and every lane then owns 32 CONSECUTIVE samples in registers.
//   Each biquad stage is solved exactly in three steps:
//     (a) per lane: FIR part + zero-state recursion over its 32 samples;
//     (b) across lanes: the 2-vector recursion  F_c = z_c + P F_{c-1}  (P = companion^32)
//         by a DPP wave scan (row_shr 1/2/4/8, row_bcast15, row_bcast31) with powers of P;
//     (c) per lane: re-run the recursion from the true entering state.
//   The state entering a super-block is carried in registers from the previous one, so a
//   segment is filtered exactly; a segment that does not start at sample 0 starts `warm`
//   samples early from zero state, `warm` chosen by the host so that the largest pole
//   radius^warm < 1e-9 (the discarded transient is below float32 resolution).
//   Energy: per lane sum of y^2, reduced across the wave with xor-shuffles in f64, split at
//   100 ms hop boundaries; 400 ms block energy = sum of 4 hop energies (K = m*S).
// Kernel 2  lufs_gate: one wave per item; float64 gating exactly as the reference
//   (z in f32, l/Gamma in f64).
#include "at_common.h"
#include <stdlib.h>

namespace {

constexpr int CHUNK = 32;            // samples per lane
constexpr int SB = 64 * CHUNK;       // samples per super-block
constexpr int ROWF = CHUNK + 4;      // padded LDS row (floats): 144 B
constexpr int MAX_STAGE = 4;
#ifndef AT_LUFS_WPS
#define AT_LUFS_WPS 3   // waves per SIMD the register allocator must allow
#endif

struct Stage {
  float b0, b1, b2, a1, a2, g;
};

struct LufsArgs {
  const float* x;      // (rows, T)
  double* E;           // (rows, H) hop energies, pre-zeroed
  float* y;            // optional (rows, T) filtered signal (general block path) or null
  int64_t T;
  int64_t rows;
  int S;               // hop (samples)
  int H;               // hops per row in E
  int H_data;          // hops that contain samples: ceil(T/S)
  int seg_hops;
  int segs_per_row;
  int warm;            // warm-up samples for segments that start after sample 0
  int nstage;
  int vec4;
  int debug;
  Stage st[MAX_STAGE];
  // P^(2^i), i = 0..6, per stage, row-major 2x2: wave-uniform, read through s_load
  double Pp[MAX_STAGE][7][4];   // float64: used once per wave to build the per-lane matrices
  float Pf[MAX_STAGE][4][4];    // float32 P^1, P^2, P^4, P^8 for the in-row scan steps
};

struct M2 {  // 2x2 double matrix
  double a, b, c, d;
};
__device__ __forceinline__ M2 mmul(const M2& x, const M2& y) {
  return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
struct M2f {  // 2x2 float matrix
  float a, b, c, d;
};

// DPP data movement (gfx9): all VALU-rate, no LDS round trip
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {  // lanes without a source read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int N> __device__ __forceinline__ float row_shr(float v) { return dpp0<0x110 | N, 0xf>(v); }  // lane i <- i-N in its 16-row
__device__ __forceinline__ float row_bcast15(float v) { return dpp0<0x142, 0xa>(v); }  // rows 1,3 <- lane 15 / 47
__device__ __forceinline__ float row_bcast31(float v) { return dpp0<0x143, 0xc>(v); }  // rows 2,3 <- lane 31
__device__ __forceinline__ float wave_shr1(float v, float lane0) {                     // lane i <- i-1, lane 0 <- lane0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0), __builtin_bit_cast(int, v),
                                                                0x138, 0xf, 0xf, false));
}
// wave64 sum with DPP only (quad_perm xor 1/2, row_half_mirror, row_mirror, row_bcast15/31);
// the total ends up in lane 63 and is broadcast with v_readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false)); // row_bcast15 -> rows 1,3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false)); // row_bcast31 -> rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct LaneMats {  // per-lane float32 powers of P = companion^CHUNK
  M2f q16;  // P^((lane & 15) + 1): folds in the previous 16-lane row
  M2f q32;  // P^((lane & 31) + 1): folds in lanes 0..31
  M2f q64;  // P^(lane + 1):        folds in the state entering the super-block
};

// Solve one biquad stage in place on the lane's CHUNK samples.
// in:  v[] = stage input; (hx1,hx2) = the two input samples preceding the lane's chunk
// io:  (ky1,ky2) = output state entering the SUPER-BLOCK (wave-uniform) in the basis (y1, y1-y2);
//      updated to the state leaving it.
template <int SI>
__device__ __forceinline__ void biquad_stage(float (&v)[CHUNK], const LufsArgs& A, float hx1, float hx2, float& ky1,
                                             float& ky2, const LaneMats& Q, int lane) {
  const Stage s = A.st[SI];
  // (a) FIR part in place + zero-state recursion
  float x1 = hx1, x2 = hx2;
  float y1 = 0.f, y2 = 0.f;
#pragma unroll
  for (int i = 0; i < CHUNK; ++i) {
    const float xn = v[i];
    float o = s.b2 * x2;
    o = fmaf(s.b1, x1, o);
    o = fmaf(s.b0, xn, o);
    v[i] = o;
    x2 = x1;
    x1 = xn;
    const float y = fmaf(-s.a1, y1, fmaf(-s.a2, y2, o));
    y2 = y1;
    y1 = y;
  }
  // (b) wave scan of the state recursion F_c = z_c + P F_{c-1}:
  //     Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8 with P, P^2, P^4, P^8),
  //     then row_bcast15 / row_bcast31 fold the previous rows in with per-lane powers of P.
  float g1 = y1, g2 = y1 - y2;  // state in the basis (y1, y1 - y2)
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    float u1, u2;
    if (step == 0) { u1 = row_shr<1>(g1); u2 = row_shr<1>(g2); }
    if (step == 1) { u1 = row_shr<2>(g1); u2 = row_shr<2>(g2); }
    if (step == 2) { u1 = row_shr<4>(g1); u2 = row_shr<4>(g2); }
    if (step == 3) { u1 = row_shr<8>(g1); u2 = row_shr<8>(g2); }
    const float n1 = fmaf(A.Pf[SI][step][0], u1, fmaf(A.Pf[SI][step][1], u2, g1));
    const float n2 = fmaf(A.Pf[SI][step][2], u1, fmaf(A.Pf[SI][step][3], u2, g2));
    g1 = n1; g2 = n2;
  }
  {
    const float u1 = row_bcast15(g1), u2 = row_bcast15(g2);
    const float n1 = fmaf(Q.q16.a, u1, fmaf(Q.q16.b, u2, g1));
    const float n2 = fmaf(Q.q16.c, u1, fmaf(Q.q16.d, u2, g2));
    g1 = n1; g2 = n2;
  }
  {
    const float u1 = row_bcast31(g1), u2 = row_bcast31(g2);
    const float n1 = fmaf(Q.q32.a, u1, fmaf(Q.q32.b, u2, g1));
    const float n2 = fmaf(Q.q32.c, u1, fmaf(Q.q32.d, u2, g2));
    g1 = n1; g2 = n2;
  }
  // F_c = G_c + P^(c+1) K ; entering state of lane c = F_{c-1} (lane 0: K)
  const float f1 = fmaf(Q.q64.a, ky1, fmaf(Q.q64.b, ky2, g1));
  const float f2 = fmaf(Q.q64.c, ky1, fmaf(Q.q64.d, ky2, g2));
  y1 = wave_shr1(f1, ky1);
  y2 = y1 - wave_shr1(f2, ky2);   // back to (y[-1], y[-2])
  ky1 = lane63(f1);
  ky2 = lane63(f2);               // carried in the (y1, y1 - y2) basis
  // (c) true recursion from the entering state
  if (s.g == 1.0f) {  // wave-uniform; every BS.1770 weighting stage has unit pass-band gain (loudness.py:259)
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
      const float y = fmaf(-s.a1, y1, fmaf(-s.a2, y2, v[i]));
      v[i] = y;
      y2 = y1;
      y1 = y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
      const float y = fmaf(-s.a1, y1, fmaf(-s.a2, y2, v[i]));
      v[i] = s.g * y;
      y2 = y1;
      y1 = y;
    }
  }
}

template <int NS, bool VEC4, bool WRITE_Y>
__global__ __launch_bounds__(256, AT_LUFS_WPS) void kweight_hop_energy(const LufsArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 64 * ROWF];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t row = wid / A.segs_per_row;
  if (row >= A.rows) return;
  const int seg = (int)(wid % A.segs_per_row);
  float* slab = lds + wave * 64 * ROWF;

  const int h0 = seg * A.seg_hops;
  const int h1 = min(h0 + A.seg_hops, A.H_data);
  const int64_t n0 = (int64_t)h0 * A.S;
  const int64_t n1 = min((int64_t)h1 * A.S, A.T);
  int64_t start = n0 - A.warm;
  if (start < 0) start = 0;
  start &= ~(int64_t)3;
  const float* __restrict__ xr = A.x + row * A.T;

  // ---- per-lane scan matrices (built once per wave in f64 from the kernarg powers of P)
  LaneMats Q[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    auto power = [&](int e) {
      M2 acc = {1.0, 0.0, 0.0, 1.0};
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (e & (1 << i)) {
          const M2 p = {A.Pp[s][i][0], A.Pp[s][i][1], A.Pp[s][i][2], A.Pp[s][i][3]};
          acc = mmul(acc, p);
        }
      }
      return M2f{(float)acc.a, (float)acc.b, (float)acc.c, (float)acc.d};
    };
    Q[s].q16 = power((lane & 15) + 1);
    Q[s].q32 = power((lane & 31) + 1);
    Q[s].q64 = power(lane + 1);
  }

  float ky1[NS], ky2[NS];  // output state entering the super-block, per stage
  float kx1[NS], kx2[NS];  // input history entering the super-block, per stage
#pragma unroll
  for (int s = 0; s < NS; ++s) { ky1[s] = ky2[s] = 0.f; kx1[s] = kx2[s] = 0.f; }

  int h_cur = -1;
  int h_run = h0;      // hop containing the first sample whose energy is counted (n0 = h0 S)
  double acc = 0.0;

  // raw tile of one super-block in registers: 8 x float4 (VEC4) or 32 x float (scalar path)
  float4 R[8];
  auto load_tile = [&](int64_t sb) __attribute__((always_inline)) {
    // Branch-free: a conditional load makes the compiler drain vmcnt before every load of the
    // tile (WAW on the destination registers) and serialises the 8 loads.  Addresses past the
    // row end are clamped into the row and the result is replaced by 0 with a select.
    if constexpr (VEC4) {  // T % 4 == 0 and sb % 4 == 0: a float4 is entirely inside or outside the row
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t g = sb + 4 * (int64_t)(i * 64 + lane);
        const int64_t gc = g < A.T ? g : A.T - 4;
        const float4 val = *reinterpret_cast<const float4*>(xr + gc);
        R[i] = g < A.T ? val : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t g = sb + (4 * i + j) * 64 + lane;
          const float val = xr[g < A.T ? g : A.T - 1];
          e[j] = g < A.T ? val : 0.f;
        }
        R[i] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  };
  load_tile(start);

  for (int64_t sb = start; sb < n1; sb += SB) {
    // ---- transpose the 2048-sample tile through LDS: lane c ends up with samples 32c..32c+31
    if constexpr (VEC4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = i * 64 + lane;
        *reinterpret_cast<float4*>(slab + (p >> 3) * ROWF + (p & 7) * 4) = R[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float e[4] = {R[i].x, R[i].y, R[i].z, R[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = (4 * i + j) * 64 + lane;
          slab[(idx >> 5) * ROWF + (idx & 31)] = e[j];
        }
      }
    }
    at::wave_sync();
    float v[CHUNK];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 q = *reinterpret_cast<const float4*>(slab + lane * ROWF + 4 * i);
      v[4 * i + 0] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
    at::wave_sync();
    // prefetch the next tile now: its HBM latency hides behind the ~700 VALU ops below
    if (sb + SB < n1) load_tile(sb + SB);

    // ---- biquad cascade
#define AT_STAGE(SI)                                                     \
    if constexpr (SI < NS) if (!A.debug) {                                             \
      const float hx1 = wave_shr1(v[CHUNK - 1], kx1[SI]);                \
      const float hx2 = wave_shr1(v[CHUNK - 2], kx2[SI]);                \
      kx1[SI] = lane63(v[CHUNK - 1]);                                    \
      kx2[SI] = lane63(v[CHUNK - 2]);                                    \
      biquad_stage<SI>(v, A, hx1, hx2, ky1[SI], ky2[SI], Q[SI], lane);   \
    }
    AT_STAGE(0) AT_STAGE(1) AT_STAGE(2) AT_STAGE(3)
#undef AT_STAGE

    const int64_t base = sb + (int64_t)lane * CHUNK;
    if constexpr (WRITE_Y) {
      // general-block path: hand the filtered signal to block_energy_kernel
      float* yr = A.y + row * A.T;
#pragma unroll
      for (int i = 0; i < CHUNK; ++i) {
        const int64_t g = base + i;
        if (g >= n0 && g < n1) yr[g] = v[i];
      }
    } else {
      // ---- hop energies
      const int64_t lo = max(sb, n0), hi = min(sb + (int64_t)SB, n1);
      if (lo < hi) {
        // hops [hf, hl] overlap [lo, hi).  lo never decreases from one super-block to the next, so the
        // hop that contains it is tracked incrementally (h_run starts at the segment's first hop):
        // the two 64-bit divisions lo / S and (hi - 1) / S cost ~15 % of the loop's instructions.
        while ((int64_t)(h_run + 1) * A.S <= lo) ++h_run;
        const int hf = h_run;
        for (int h = hf; (int64_t)h * A.S < hi; ++h) {
          const int64_t a = max(lo, (int64_t)h * A.S), b = min(hi, (int64_t)(h + 1) * A.S);
          float e = 0.f;
          if (a == sb && b == sb + SB) {
#pragma unroll
            for (int i = 0; i < CHUNK; ++i) e = fmaf(v[i], v[i], e);
          } else {
            const int ilo = (int)max((int64_t)0, min((int64_t)CHUNK, a - base));
            const int ihi = (int)max((int64_t)0, min((int64_t)CHUNK, b - base));
#pragma unroll
            for (int i = 0; i < CHUNK; ++i) {
              const float m = (i >= ilo && i < ihi) ? v[i] : 0.f;
              e = fmaf(m, m, e);
            }
          }
          const double tot = (double)wave_sum_dpp(e);
          if (h != h_cur) {
            if (h_cur >= 0 && lane == 0) A.E[row * A.H + h_cur] = acc;
            h_cur = h;
            acc = 0.0;
          }
          acc += tot;
        }
      }
    }
  }
  if constexpr (!WRITE_Y) {
    if (h_cur >= 0 && lane == 0) A.E[row * A.H + h_cur] = acc;
  }
}

// ---- kweight_hop_energy_dma: the hop-energy path for 16-byte-aligned rows -------------------------
// Same arithmetic as kweight_hop_energy<NS, true, false> (bit-identical hop energies are NOT required,
// the per-piece sums are formed in a different order), restructured around the three things the ISA of
// that kernel spends outside the recursions (profiles/r02_notes.md, r03_notes.md):
//  * all segment / hop bookkeeping is wave-uniform and lives in SGPRs (the wave index comes from
//    readfirstlane): no 64-bit vector compares, no exec-mask branches, no division in the loop;
//  * hop energies are formed DIRECTLY per piece: every lane sums its 32 squares once; a cut (a hop
//    boundary, or the end of the data) at offset o inside lane c splits that lane's chunk into the
//    squares below and above o with wave-uniform branches per group of four samples, and the piece
//    sums are  [lanes strictly between two cuts: full sums] + [the two boundary lanes: partial sums]
//    -- 32 + ~36 FMAs per cut instead of two masked 32-sample passes with per-sample selects.  No
//    piece is ever obtained as a difference of two sums (a quiet hop behind a loud one keeps its
//    relative accuracy: it decides the -70 LUFS gate);
//  * the next tile is prefetched by LDS-DMA (global_load_lds_dwordx4) straight into the wave's 8 KB
//    tile instead of 8 float4 staging registers: 32 VGPRs fewer, 4 waves per SIMD instead of 3.
//    DMA piece i / lane l writes LDS slot 64 i + l (fixed by the hardware); the GLOBAL chunk each lane
//    fetches is permuted inside its 128-byte line (k = (l & 7) ^ ((c >> 1) & 7)), so HBM access stays
//    line-coalesced and lane c's k-th float4 sits at slot 8 c + (k ^ ((c >> 1) & 7)): the 16 lanes a
//    ds_read_b128 serves per cycle hit 16 distinct bank groups (a plain layout is 8-way conflicted).
#ifndef AT_LUFS_WPS_DMA
#define AT_LUFS_WPS_DMA 4
#endif

// squares of the lane's chunk below / at-or-above the wave-uniform offset o (0..31)
__device__ __forceinline__ void split_energy(const float (&v)[CHUNK], int o, float& elo, float& ehi) {
  float lo = 0.f, hi = 0.f;
#pragma unroll
  for (int g = 0; g < CHUNK / 4; ++g) {
    if (o >= 4 * g + 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) lo = fmaf(v[4 * g + j], v[4 * g + j], lo);
    } else if (o <= 4 * g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) hi = fmaf(v[4 * g + j], v[4 * g + j], hi);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sq = v[4 * g + j] * v[4 * g + j];
        if (4 * g + j < o) lo += sq; else hi += sq;
      }
    }
  }
  elo = lo; ehi = hi;
}

// One biquad stage as biquad_stage<SI>(), with the per-lane scan matrices read from the block's LDS
// table (3 float4 per lane: q16, q32, q64, then the four in-row step matrices) instead of 12 VGPRs and
// 16 SGPRs per stage (the SGPR file was over-subscribed: 30 spills reloaded with v_readlane per tile).
template <int SI>
__device__ __forceinline__ void biquad_stage_q(float (&v)[CHUNK], const LufsArgs& A, float hx1, float hx2, float& ky1,
                                               float& ky2, const float4* __restrict__ q /* [3][64] per-lane + [4] uniform */, int lane) {
  const Stage s = A.st[SI];
  float x1 = hx1, x2 = hx2;
  float y1 = 0.f, y2 = 0.f;
#pragma unroll
  for (int i = 0; i < CHUNK; ++i) {
    const float xn = v[i];
    float o = s.b2 * x2;
    o = fmaf(s.b1, x1, o);
    o = fmaf(s.b0, xn, o);
    v[i] = o;
    x2 = x1;
    x1 = xn;
    const float y = fmaf(-s.a1, y1, fmaf(-s.a2, y2, o));
    y2 = y1;
    y1 = y;
  }
  const float4 q16 = q[lane], q32 = q[64 + lane], q64 = q[128 + lane];
  float g1 = y1, g2 = y1 - y2;
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    float u1, u2;
    if (step == 0) { u1 = row_shr<1>(g1); u2 = row_shr<1>(g2); }
    if (step == 1) { u1 = row_shr<2>(g1); u2 = row_shr<2>(g2); }
    if (step == 2) { u1 = row_shr<4>(g1); u2 = row_shr<4>(g2); }
    if (step == 3) { u1 = row_shr<8>(g1); u2 = row_shr<8>(g2); }
    const float4 pf = q[192 + step];   // P^(2^step), the same for every lane (LDS broadcast read)
    const float n1 = fmaf(pf.x, u1, fmaf(pf.y, u2, g1));
    const float n2 = fmaf(pf.z, u1, fmaf(pf.w, u2, g2));
    g1 = n1; g2 = n2;
  }
  {
    const float u1 = row_bcast15(g1), u2 = row_bcast15(g2);
    const float n1 = fmaf(q16.x, u1, fmaf(q16.y, u2, g1));
    const float n2 = fmaf(q16.z, u1, fmaf(q16.w, u2, g2));
    g1 = n1; g2 = n2;
  }
  {
    const float u1 = row_bcast31(g1), u2 = row_bcast31(g2);
    const float n1 = fmaf(q32.x, u1, fmaf(q32.y, u2, g1));
    const float n2 = fmaf(q32.z, u1, fmaf(q32.w, u2, g2));
    g1 = n1; g2 = n2;
  }
  const float f1 = fmaf(q64.x, ky1, fmaf(q64.y, ky2, g1));
  const float f2 = fmaf(q64.z, ky1, fmaf(q64.w, ky2, g2));
  y1 = wave_shr1(f1, ky1);
  y2 = y1 - wave_shr1(f2, ky2);
  ky1 = lane63(f1);
  ky2 = lane63(f2);
  if (s.g == 1.0f) {
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
      const float y = fmaf(-s.a1, y1, fmaf(-s.a2, y2, v[i]));
      v[i] = y;
      y2 = y1;
      y1 = y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
      const float y = fmaf(-s.a1, y1, fmaf(-s.a2, y2, v[i]));
      v[i] = s.g * y;
      y2 = y1;
      y1 = y;
    }
  }
}

// Sample indices are 32-bit here (the launcher routes rows of 2^31 - 2^22 samples or more to the
// register-staged kernel).
template <int NS>
__global__ __launch_bounds__(256, (NS <= 2 ? AT_LUFS_WPS_DMA : 3)) void kweight_hop_energy_dma(const LufsArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[4 * SB];
  __shared__ __attribute__((aligned(16))) float4 qtab[NS][3 * 64 + 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  if (wid >= A.rows * (int64_t)A.segs_per_row) return;   // wave-uniform
  const int64_t row = wid / A.segs_per_row;
  const int seg = (int)(wid - row * A.segs_per_row);
  float* tile = lds + wave * SB;

  const int T = (int)A.T, S = A.S;
  const int h0 = seg * A.seg_hops;
  const int h1 = min(h0 + A.seg_hops, A.H_data);
  const int n0 = h0 * S;
  const int n1 = (int)min((int64_t)h1 * S, A.T);
  int start = n0 - A.warm;
  if (start < 0) start = 0;
  start &= ~3;
  const float* __restrict__ xr = A.x + row * A.T;
  double* __restrict__ Erow = A.E + row * A.H;

  // per-lane scan matrices -> LDS.  Every wave of the block writes the same values (no block barrier:
  // waves past the end of the grid have already left).
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    auto power = [&](int e) {
      M2 acc = {1.0, 0.0, 0.0, 1.0};
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (e & (1 << i)) {
          const M2 p = {A.Pp[s][i][0], A.Pp[s][i][1], A.Pp[s][i][2], A.Pp[s][i][3]};
          acc = mmul(acc, p);
        }
      }
      return make_float4((float)acc.a, (float)acc.b, (float)acc.c, (float)acc.d);
    };
    qtab[s][lane] = power((lane & 15) + 1);
    qtab[s][64 + lane] = power((lane & 31) + 1);
    qtab[s][128 + lane] = power(lane + 1);
    if (lane < 4) qtab[s][192 + lane] = make_float4(A.Pf[s][lane][0], A.Pf[s][lane][1], A.Pf[s][lane][2], A.Pf[s][lane][3]);
  }
  at::wave_sync();
  float ky1[NS], ky2[NS], kx1[NS], kx2[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { ky1[s] = ky2[s] = 0.f; kx1[s] = kx2[s] = 0.f; }

  // DMA: piece i / lane l lands in LDS slot 64 i + l and fetches the chunk k = (l & 7) ^ ((c >> 1) & 7) of
  // lane-row c = 8 i + (l >> 3), i.e. float offset 256 i + [32 (l >> 3) + 4 k0] ^ (16 (i & 1)) with
  // k0 = (l & 7) ^ (l >> 4): two per-lane offsets, the piece index goes into the instruction offset
  // (which moves the global AND the LDS address).
  const int offA = 32 * (lane >> 3) + 4 * ((lane & 7) ^ (lane >> 4));
  const int offB = offA ^ 16;
  const int rd_base = lane * CHUNK + 4 * ((lane >> 1) & 7);   // the reader's float4 k sits at rd_base ^ (4 k)
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_tile = [&](int sb) __attribute__((always_inline)) {
    if (sb + SB <= T) {                                     // wave-uniform
      const float* gA = xr + sb + offA;
      const float* gB = xr + sb + offB;
      __builtin_amdgcn_global_load_lds((gptr_t)gA, (lptr_t)tile, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gB, (lptr_t)tile, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gA, (lptr_t)tile, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gB, (lptr_t)tile, 16, 3072, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gA + 1024), (lptr_t)(tile + 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gB + 1024), (lptr_t)(tile + 1024), 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gA + 1024), (lptr_t)(tile + 1024), 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gB + 1024), (lptr_t)(tile + 1024), 16, 3072, 0);
    } else {                                                // last tile of a row: chunk addresses clamped into the row
#pragma unroll 1
      for (int i = 0; i < 8; ++i) {
        int off = sb + 256 * i + ((i & 1) ? offB : offA);
        off = off > T - 4 ? T - 4 : off;
        __builtin_amdgcn_global_load_lds((gptr_t)(xr + off), (lptr_t)(tile + 256 * i), 16, 0, 0);
      }
    }
  };

  // hop tracking: h = hop that contains the current position, hb = first sample of hop h + 1
  int h = start / S;
  int hb = (h + 1) * S;                  // < T + S: the launcher keeps T below 2^31 - 2^22
  double acc = 0.0;
  bool acc_valid = false;

  issue_tile(start);
  for (int sb = start; sb < n1; sb += SB) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the tile has landed (LDS-DMA is covered by vmcnt only)
    at::wave_sync();
    float v[CHUNK];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 q = *reinterpret_cast<const float4*>(tile + (rd_base ^ (4 * k)));
      v[4 * k + 0] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // values are in registers before the tile is overwritten
    at::wave_sync();
    if (sb + SB < n1) issue_tile(sb + SB);
    if (sb + SB > T) {                                       // last tile of the row: samples past the end are zeros
      const int base = sb + lane * CHUNK;
#pragma unroll
      for (int i = 0; i < CHUNK; ++i) v[i] = base + i < T ? v[i] : 0.f;
    }

#define AT_STAGE(SI)                                                     \
    if constexpr (SI < NS) {                                             \
      const float hx1 = wave_shr1(v[CHUNK - 1], kx1[SI]);                \
      const float hx2 = wave_shr1(v[CHUNK - 2], kx2[SI]);                \
      kx1[SI] = lane63(v[CHUNK - 1]);                                    \
      kx2[SI] = lane63(v[CHUNK - 2]);                                    \
      biquad_stage_q<SI>(v, A, hx1, hx2, ky1[SI], ky2[SI], &qtab[SI][0], lane);   \
    }
    AT_STAGE(0) AT_STAGE(1) AT_STAGE(2) AT_STAGE(3)
#undef AT_STAGE

    // ---- hop energies of this super-block, piece by piece (all control flow is wave-uniform)
    float e_full = 0.f;
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) e_full = fmaf(v[i], v[i], e_full);
    const int sb_end = sb + SB;
    int c_prev = -1, o_prev = 0;
    float carry = 0.f;
    while (true) {
      const int cut = hb < n1 ? hb : n1;
      if (cut >= sb_end) {                                   // the piece runs to the end of the block
        if (h >= h0) {
          const float contrib = (lane > c_prev ? e_full : 0.f) + carry;
          acc += (double)wave_sum_dpp(contrib);
          acc_valid = true;
        }
        break;
      }
      const int p = cut - sb;
      const int c = p >> 5, o = p & 31;
      float elo, ehi;
      split_energy(v, o, elo, ehi);
      if (h >= h0) {
        float contrib;
        if (c != c_prev) {
          contrib = carry + ((lane > c_prev && lane < c) ? e_full : 0.f) + (lane == c ? elo : 0.f);
        } else {
          // two cuts inside one lane's chunk: only the end of the data can follow a hop boundary this
          // closely (hop boundaries are >= 100 ms apart).  Sum the samples [o_prev, o) of that lane directly.
          float mid = 0.f;
#pragma unroll
          for (int i = 0; i < CHUNK; ++i) mid += (i >= o_prev && i < o) ? v[i] * v[i] : 0.f;
          contrib = lane == c ? mid : 0.f;
        }
        acc += (double)wave_sum_dpp(contrib);
        acc_valid = true;
      }
      if (cut == n1) break;                                  // end of this segment's data (the outer loop ends too)
      if (acc_valid) {
        if (lane == 0) Erow[h] = acc;
        acc = 0.0;
        acc_valid = false;
      }
      ++h;
      hb += S;
      carry = lane == c ? ehi : 0.f;
      c_prev = c;
      o_prev = o;
    }
  }
  if (acc_valid && lane == 0) Erow[h] = acc;
}

// General block energies from a filtered signal (K not a multiple of S): one wave per (row, block).
__global__ __launch_bounds__(256) void block_energy_kernel(const float* __restrict__ y, double* __restrict__ Z,
                                                            int64_t rows, int64_t T, int K, int S, int nblk) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= rows * nblk) return;
  const int64_t row = wid / nblk;
  const int j = (int)(wid % nblk);
  const int64_t lo = (int64_t)j * S, hi = min(lo + (int64_t)K, T);
  const float* yr = y + row * T;
  double acc = 0.0;
  for (int64_t n = lo + lane; n < hi; n += 64) {
    const float q = yr[n];
    acc += (double)q * (double)q;
  }
  acc = at::wave_sum(acc);
  if (lane == 0) Z[wid] = acc;
}

struct GateArgs {
  const double* E;   // (B*C, H) hop energies   (hops_per_block > 0)  or (B*C, nblk) block energies (== 0)
  float* out;        // (B)
  int B, C, H, nblk, hops_per_block;
  float inv_norm;    // f32(1/(T_g*rate))
  float floor_db;    // clamp (NaN = none)
};

__device__ __forceinline__ float block_z(const GateArgs& A, int b, int c, int j) {
  const double* e = A.E + ((int64_t)b * A.C + c) * A.H;
  double s;
  if (A.hops_per_block > 0) {
    s = 0.0;
    for (int i = 0; i < A.hops_per_block; ++i) s += e[j + i];
  } else {
    s = e[j];
  }
  return (float)s * A.inv_norm;  // loudness.py:214, float32
}

__global__ __launch_bounds__(64) void lufs_gate(const GateArgs A) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const double Gc[5] = {1.0, 1.0, 1.0, 1.41, 1.41};  // loudness.py:49
  const double GAMMA_A = -70.0;

  // pass 1: absolute gate
  double sum1[5] = {0, 0, 0, 0, 0};
  double cnt1 = 0.0;
  for (int j = lane; j < A.nblk; j += 64) {
    double tot = 0.0;
    float z[5];
    for (int c = 0; c < A.C; ++c) { z[c] = block_z(A, b, c, j); tot += Gc[c] * (double)z[c]; }
    const double l = -0.691 + 10.0 * log10(tot);
    if (l > GAMMA_A) {
      cnt1 += 1.0;
      for (int c = 0; c < A.C; ++c) sum1[c] += (double)z[c];
    }
  }
  cnt1 = at::wave_sum(cnt1);
  double tot_r = 0.0;
  for (int c = 0; c < A.C; ++c) {
    const float zavg = (float)at::wave_sum(sum1[c]) / (float)cnt1;  // 0/0 -> NaN as in the reference
    tot_r += (double)zavg * Gc[c];
  }
  const double gamma_r = -0.691 + 10.0 * log10(tot_r) - 10.0;

  // pass 2: absolute + relative gate
  double sum2[5] = {0, 0, 0, 0, 0};
  double cnt2 = 0.0;
  for (int j = lane; j < A.nblk; j += 64) {
    double tot = 0.0;
    float z[5];
    for (int c = 0; c < A.C; ++c) { z[c] = block_z(A, b, c, j); tot += Gc[c] * (double)z[c]; }
    const double l = -0.691 + 10.0 * log10(tot);
    if (l > GAMMA_A && l > gamma_r) {  // NaN gamma_r -> nothing passes, as in the reference
      cnt2 += 1.0;
      for (int c = 0; c < A.C; ++c) sum2[c] += (double)z[c];
    }
  }
  cnt2 = at::wave_sum(cnt2);
  double tot_f = 0.0;
  for (int c = 0; c < A.C; ++c) {
    float zavg = (float)at::wave_sum(sum2[c]) / (float)cnt2;
    if (isnan(zavg)) zavg = 0.f;
    if (isinf(zavg)) zavg = zavg > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    tot_f += Gc[c] * (double)zavg;
  }
  float lufs = (float)(-0.691 + 10.0 * log10(tot_f));
  if (!isnan(A.floor_db)) lufs = fmaxf(lufs, A.floor_db);
  if (lane == 0) A.out[b] = lufs;
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

// Workspace layout: [E: rows*H doubles][y: rows*T floats, only when K % S != 0]
int64_t at_lufs_workspace_bytes(int64_t B, int64_t C, int64_t T, int K, int S) {
  if (B < 0 || C <= 0 || T <= 0 || K <= 0 || S <= 0) return AT_ERR_INVALID;
  const int64_t rows = B * C;
  const int64_t Tk = T > K ? T : K;
  const int64_t nblk = (Tk - K + S - 1) / S + 1;
  if (K % S == 0) {
    const int64_t H = nblk - 1 + K / S;
    return align_up(rows * H * 8, 256);
  }
  return align_up(rows * nblk * 8, 256) + align_up(rows * T * 4, 256);
}

// sos: nstage x 6 doubles (b0,b1,b2,a0,a1,a2) as designed; gains: nstage passband gains.
// Coefficients are rounded to float32 exactly as loudness.py:118-119 does (.float()).
int at_lufs_f32(const float* x, int64_t B, int64_t C, int64_t T, const double* sos, const double* gains, int nstage,
                int K, int S, double inv_norm, float floor_db, int warm, float* out, void* workspace,
                int64_t workspace_bytes, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !sos || !gains || !out || B < 0 || C <= 0 || C > 5 || T <= 0 || nstage <= 0 || nstage > MAX_STAGE ||
      K <= 0 || S <= 0 || warm < 0)
    return AT_ERR_INVALID;
  if (B == 0) return AT_OK;
  const int64_t need = at_lufs_workspace_bytes(B, C, T, K, S);
  if (!workspace || workspace_bytes < need) return AT_ERR_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = B * C;
  const int64_t Tk = T > K ? T : K;
  const int nblk = (int)((Tk - K + S - 1) / S + 1);
  const bool hop_path = (K % S == 0);

  LufsArgs A;
  A.x = x; A.T = T; A.rows = rows; A.S = S; A.nstage = nstage; A.warm = warm;
  for (int s = 0; s < nstage; ++s) {
    const float a0 = (float)sos[6 * s + 3];
    A.st[s].b0 = (float)sos[6 * s + 0] / a0;
    A.st[s].b1 = (float)sos[6 * s + 1] / a0;
    A.st[s].b2 = (float)sos[6 * s + 2] / a0;
    A.st[s].a1 = (float)sos[6 * s + 4] / a0;
    A.st[s].a2 = (float)sos[6 * s + 5] / a0;
    A.st[s].g = (float)gains[s];
    // P = companion^CHUNK of the float32-rounded recursion, then repeated squares
    const double a1 = (double)A.st[s].a1, a2 = (double)A.st[s].a2;
    double p1 = 1.0, p2 = 0.0, q1 = 0.0, q2 = 1.0;
    for (int i = 0; i < CHUNK; ++i) {
      const double pn = -a1 * p1 - a2 * p2; p2 = p1; p1 = pn;
      const double qn = -a1 * q1 - a2 * q2; q2 = q1; q1 = qn;
    }
    double P[4] = {p1, q1, p2, q2};  // (y[31],y[30]) = P (y[-1],y[-2])
    for (int i = 0; i < 7; ++i) {
      // stored in the basis (y1, y1 - y2): P' = T P T with T = T^-1 = [[1,0],[1,-1]].  A filter with
      // poles next to z = 1 (the 38 Hz high-pass) has |entries of P| ~ 30 with opposite signs
      // acting on two nearly equal states; in this basis the entries are O(1) and float32 is enough.
      const double Pt[4] = {P[0] + P[1], -P[1], P[0] - P[2] + P[1] - P[3], -(P[1] - P[3])};
      for (int e = 0; e < 4; ++e) { A.Pp[s][i][e] = Pt[e]; if (i < 4) A.Pf[s][i][e] = (float)Pt[e]; }
      const double n[4] = {P[0] * P[0] + P[1] * P[2], P[0] * P[1] + P[1] * P[3], P[2] * P[0] + P[3] * P[2],
                           P[2] * P[1] + P[3] * P[3]};
      for (int e = 0; e < 4; ++e) P[e] = n[e];
    }
  }
  static const int dbg_mode = at::env_int_once("AT_LUFS_DEBUG", 0);
  A.debug = dbg_mode;
  A.vec4 = ((T % 4) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0) ? 1 : 0;
  double* E = reinterpret_cast<double*>(workspace);
  int H;
  if (hop_path) {
    H = nblk - 1 + K / S;
    A.y = nullptr;
  } else {
    H = nblk;
    A.y = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align_up(rows * (int64_t)nblk * 8, 256));
  }
  A.E = E; A.H = H;
  // kernel choice: the LDS-DMA kernel needs 16-byte aligned rows and the hop path (AT_LUFS_KERNEL=1 forces the
  // register-staged kernel: A/B measurements, tools/lufsab.py; re-read per call under AT_LUFS_TUNE=1)
  static const int tune_each_call = at::env_int_once("AT_LUFS_TUNE", 0);
  static int kernel_sel = at::env_int_once("AT_LUFS_KERNEL", 0);
  if (tune_each_call) kernel_sel = at::env_int_once("AT_LUFS_KERNEL", 0);
  const bool use_dma = hop_path && A.vec4 && kernel_sel != 1 && T >= 4 && T < (1LL << 31) - (1LL << 22) && S < (1 << 22);
  // segmentation in units of S samples ("hops") even on the general path
  const int64_t H_data = (T + S - 1) / S;
  A.H_data = (int)H_data;
  // Segments per row: minimise  rounds x (segment + warm-up)  where a round is one wave on every
  // resident slot (12 per CU at 3 waves/SIMD).  Wave quantisation matters: 3200 waves on 3072
  // slots run two rounds for the work of 1.04 (B=64 took 2x its share of the B=512 time).
  int64_t seg;
  {
    const int n_cu = at::device_cu_count();
    // resident waves per SIMD: the LDS-DMA kernel holds four with one or two filter stages, three with the per-lane scan
    // matrices of three or four stages in LDS (Fenton/Lee, Dash: the hint of four was not met there and the segmentation
    // below planned for a quarter more slots than exist -- a second, nearly empty round)
    const int64_t slots = (int64_t)n_cu * 4 * (use_dma ? (nstage <= 2 ? AT_LUFS_WPS_DMA : 3) : AT_LUFS_WPS);
    int64_t best_s = 1;
    double best_cost = 1e300;
    // Cost of a segmentation (round 6, sessions s14 / s15): a wave's work (segment + warm-up) times what its SIMD's residents cost
    // it.  Until round 6 the model was rounds x work -- every resident wave at full speed up to the slot count -- and picked as
    // many segments as one round holds: 3200 waves of 21.8 k samples at 64 items, where 1920 waves of 35 k run 10 % faster.
    // Measured per-wave time / work against the waves q on the busiest SIMD (B = 64, 128 rows, nine segmentations):
    // 1.24 / 1.9 / 2.6 / 3.5 for q = 1 .. 4 -- a SIMD's throughput is 0.81 / 1.05 / 1.15 / 1.15 of q = 1's: two residents nearly
    // saturate it, a lone wave is latency-bound.
    static const double G_OF_Q[5] = {0.0, 1.24, 1.9, 2.6, 3.5};
    const int64_t n_simd = (int64_t)n_cu * 4;
    const int64_t wps = slots / n_simd;                            // resident waves per SIMD (3 or 4)
    for (int64_t sp = 1; sp <= H_data && sp <= 4096; ++sp) {
      const int64_t sh = (H_data + sp - 1) / sp;                  // hops per segment
      if ((H_data + sh - 1) / sh != sp) continue;                 // not a distinct segmentation
      const int64_t q = (rows * sp + n_simd - 1) / n_simd;        // waves the busiest SIMD runs
      const double g = q <= wps ? G_OF_Q[q] : G_OF_Q[wps] * (double)q / (double)wps;
      const double cost = g * ((double)sh * S + (sp > 1 ? (double)warm : 0.0));
      if (cost < best_cost * 0.999) { best_cost = cost; best_s = sp; }
    }
    seg = (H_data + best_s - 1) / best_s;
    static const int want_waves = at::env_int_once("AT_LUFS_WAVES", 0);  // development override: target wave count
    if (want_waves > 0) seg = (rows * H_data + want_waves - 1) / want_waves;
  }
  if (seg < 1) seg = 1;
  if (seg > H_data) seg = H_data;
  A.seg_hops = (int)seg;
  A.segs_per_row = (int)((H_data + seg - 1) / seg);

  // Every hop that holds samples (h < H_data) is stored by exactly one wave; the table only needs zeros where no wave
  // writes: the hops behind the data of a signal shorter than one gating block (T < K); the register-staged kernel and
  // the general path (K not a multiple of S) keep the memset.
  // (The unconditional memset was a 4 us launch per call: 1 % of the 64-item share of an 8-GPU run.)
  if (!use_dma || H > H_data) {
    hipError_t e = hipMemsetAsync(E, 0, (size_t)rows * H * 8, st);
    if (e != hipSuccess) return AT_ERR_HIP(e);
  }
  const int64_t waves = rows * A.segs_per_row;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define AT_LAUNCH_NS(NSV)                                                                          \
  case NSV:                                                                                        \
    if (hop_path) {                                                                                \
      if (A.vec4) hipLaunchKernelGGL((kweight_hop_energy<NSV, true, false>), grid, block, 0, st, A);  \
      else hipLaunchKernelGGL((kweight_hop_energy<NSV, false, false>), grid, block, 0, st, A);        \
    } else {                                                                                       \
      if (A.vec4) hipLaunchKernelGGL((kweight_hop_energy<NSV, true, true>), grid, block, 0, st, A);   \
      else hipLaunchKernelGGL((kweight_hop_energy<NSV, false, true>), grid, block, 0, st, A);         \
    }                                                                                              \
    break;
  if (use_dma) {
    switch (nstage) {
      case 1: hipLaunchKernelGGL((kweight_hop_energy_dma<1>), grid, block, 0, st, A); break;
      case 2: hipLaunchKernelGGL((kweight_hop_energy_dma<2>), grid, block, 0, st, A); break;
      case 3: hipLaunchKernelGGL((kweight_hop_energy_dma<3>), grid, block, 0, st, A); break;
      case 4: hipLaunchKernelGGL((kweight_hop_energy_dma<4>), grid, block, 0, st, A); break;
    }
  } else {
    switch (nstage) {
      AT_LAUNCH_NS(1) AT_LAUNCH_NS(2) AT_LAUNCH_NS(3) AT_LAUNCH_NS(4)
    }
  }
#undef AT_LAUNCH_NS
  if (!hop_path) {
    AT_LAUNCH_CHECK();
    const int64_t w2 = rows * nblk;
    hipLaunchKernelGGL(block_energy_kernel, dim3((unsigned)((w2 + 3) / 4)), block, 0, st, A.y, E, rows, T, K, S, nblk);
  }
  AT_LAUNCH_CHECK();

  GateArgs G;
  G.E = E; G.out = out; G.B = (int)B; G.C = (int)C; G.H = H; G.nblk = nblk;
  G.hops_per_block = hop_path ? K / S : 0;
  G.inv_norm = (float)inv_norm;
  G.floor_db = floor_db;
  hipLaunchKernelGGL(lufs_gate, dim3((unsigned)B), dim3(64), 0, st, G);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // extern "C"
