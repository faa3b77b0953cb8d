// Row reductions and impulse-response preparation around the FFT convolution (gfx950).
//
// The reference does these as chains of whole-tensor torch ops (abs -> max -> where -> mul ...);
// on the GPU every link of such a chain is a full HBM pass.  Each entry point here is ONE pass
// (or one read-twice/write-once pass for alter_drr) with one workgroup per (item, channel) row:
//   at_absmax_f32      audiotools/core/effects.py:100,118,160,175,213-216  x.abs().max(-1), argmax
//   at_roll_pad_f32    effects.py:85-100  zero_pad(other) + roll so the |peak| sits at sample 0
//   at_alter_drr_f32   effects.py:540-647  decompose_ir + solve_alpha + alter_drr + ensure_max_of_audio
#include "at_common.h"

namespace {

constexpr int RT = 512;  // threads per row workgroup

struct __attribute__((packed, aligned(4))) q4u { float x, y, z, w; };  // dword-aligned quad

struct MaxIdx {
  float v;
  int64_t i;
};

// first-occurrence arg-max: larger value wins, ties go to the smaller index, NaN beats every
// number (torch semantics: abs().max() / argmax() propagate NaN and report the first one)
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {
  const bool an = a.v != a.v, bn = b.v != b.v;
  if (an || bn) return (bn && (!an || b.i < a.i)) ? b : a;
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
// per-thread running maximum in index order: `cand` replaces the current best when it is larger or
// is the first NaN (a NaN best is sticky: every comparison with it is false)
__device__ __forceinline__ bool takes_over(float cand, float best) { return !(cand <= best) && best == best; }

__device__ __forceinline__ MaxIdx block_argmax(MaxIdx m, MaxIdx* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxIdx other;
    other.v = __shfl_xor(m.v, o, 64);
    other.i = __shfl_xor(m.i, o, 64);
    m = better(m, other);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  MaxIdx r = sh[0];
  for (int w = 1; w < RT / 64; ++w) r = better(r, sh[w]);
  return r;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = at::wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < RT / 64; ++w) r += sh[w];
  return r;
}

__device__ __forceinline__ float block_max(float v, float* sh) {
  v = at::wave_max(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < RT / 64; ++w) r = fmaxf(r, sh[w]);
  return r;
}

// ---- per-row max |x| and its first index ------------------------------------------------------
__global__ __launch_bounds__(RT) void absmax_kernel(const float* __restrict__ x, int64_t rows, int64_t T,
                                                    float* __restrict__ vmax, int64_t* __restrict__ imax) {
  __shared__ MaxIdx sh[RT / 64];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* __restrict__ xr = x + row * T;
    MaxIdx m{-1.0f, 0};
    const bool vec = ((reinterpret_cast<uintptr_t>(xr) & 15) == 0);
    const int64_t T4 = vec ? T / 4 : 0;
    // 4 independent 16-B loads in flight per thread (a load-then-wait loop is latency bound)
    int64_t i = threadIdx.x;
    for (; i + 3 * RT < T4; i += 4 * RT) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(xr)[i + k * RT];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a[4] = {fabsf(v[k].x), fabsf(v[k].y), fabsf(v[k].z), fabsf(v[k].w)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (takes_over(a[u], m.v)) { m.v = a[u]; m.i = 4 * (i + k * RT) + u; }
      }
    }
    for (; i < T4; i += RT) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (takes_over(a[u], m.v)) { m.v = a[u]; m.i = 4 * i + u; }
    }
    for (int64_t j = 4 * T4 + threadIdx.x; j < T; j += RT) {
      const float a = fabsf(xr[j]);
      if (takes_over(a, m.v)) { m.v = a; m.i = j; }
    }
    m = block_argmax(m, sh);
    if (threadIdx.x == 0) {
      vmax[row] = m.v;
      if (imax) imax[row] = m.i;
    }
  }
}

// ---- out[r, n] = xz[(n + shift[r]) mod T],  xz = x zero-padded (or truncated) to length T -------
__global__ __launch_bounds__(256) void roll_pad_kernel(const float* __restrict__ x, int64_t rows, int64_t L,
                                                       const int64_t* __restrict__ shift, int64_t T,
                                                       float* __restrict__ out) {
  const int64_t row = blockIdx.y;
  const float* __restrict__ xr = x + row * L;
  float* __restrict__ orow = out + row * T;
  int64_t s = shift ? shift[row] % T : 0;
  if (s < 0) s += T;
  const int64_t Lc = L < T ? L : T;
  const bool vec = (T % 4) == 0 && ((reinterpret_cast<uintptr_t>(orow) & 15) == 0);
  if (vec) {
    // 4 consecutive outputs per thread: one dword-aligned 16-B load when the 4 sources are inside
    // the row (no wrap, no zero tail), element-wise otherwise; aligned 16-B stores
    for (int64_t n = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); n < T; n += 4 * (int64_t)gridDim.x * blockDim.x) {
      int64_t src = n + s;
      if (src >= T) src -= T;
      float4 o;
      if (src + 3 < Lc) {
        const q4u v = *reinterpret_cast<const q4u*>(xr + src);
        o = make_float4(v.x, v.y, v.z, v.w);
      } else {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int64_t su = src + u;
          if (su >= T) su -= T;
          e[u] = su < Lc ? xr[su] : 0.f;
        }
        o = make_float4(e[0], e[1], e[2], e[3]);
      }
      *reinterpret_cast<float4*>(orow + n) = o;
    }
  } else {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < T; n += (int64_t)gridDim.x * blockDim.x) {
      int64_t src = n + s;
      if (src >= T) src -= T;
      orow[n] = src < Lc ? xr[src] : 0.f;
    }
  }
}

// ---- direct-to-reverberant ratio change of an impulse response ---------------------------------
// One workgroup per (item, channel) row; the row is swept three times (arg-max, energies, output),
// sweeps two and three hit L2 / Infinity Cache.
__global__ __launch_bounds__(RT) void alter_drr_kernel(const float* __restrict__ x, int64_t B, int C, int64_t T, int t0,
                                                       const float* __restrict__ drr /* (B) */,
                                                       float* __restrict__ out, float* __restrict__ vmax, int64_t* __restrict__ imax) {
  __shared__ MaxIdx shm[RT / 64];
  __shared__ float shf[RT / 64];
  for (int64_t row = blockIdx.x; row < B * C; row += gridDim.x) {
    const int64_t b = row / C;
    const float* __restrict__ xr = x + row * T;
    const float* __restrict__ x0 = x + b * C * T;  // channel 0 of the item: its early span is the window
    // sweep 1: signed arg-max of this row (and of channel 0).  Rows are read 16 B per lane with
    // two loads in flight when the row is 16-B aligned (T % 4 == 0), element-wise otherwise.
    const bool vec = (T % 4) == 0 && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    auto row_argmax = [&](const float* __restrict__ p) {
      MaxIdx mm{-INFINITY, 0};
      if (vec) {
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
        const int64_t T4 = T / 4;
        int64_t i = threadIdx.x;
        for (; i + RT < T4; i += 2 * RT) {
          const float4 a = p4[i], b = p4[i + RT];
          const float ea[4] = {a.x, a.y, a.z, a.w}, eb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) if (ea[u] > mm.v) { mm.v = ea[u]; mm.i = 4 * i + u; }
#pragma unroll
          for (int u = 0; u < 4; ++u) if (eb[u] > mm.v) { mm.v = eb[u]; mm.i = 4 * (i + RT) + u; }
        }
        for (; i < T4; i += RT) {
          const float4 a = p4[i];
          const float ea[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) if (ea[u] > mm.v) { mm.v = ea[u]; mm.i = 4 * i + u; }
        }
      } else {
        for (int64_t i = threadIdx.x; i < T; i += RT) {
          const float v = p[i];
          if (v > mm.v) { mm.v = v; mm.i = i; }
        }
      }
      return block_argmax(mm, shm);
    };
    MaxIdx m = row_argmax(xr);
    MaxIdx m0 = (xr != x0) ? row_argmax(x0) : m;
    const int64_t e_lo = m.i - t0, e_hi = m.i + t0;      // early span of this row
    const int64_t w_lo = m0.i - t0, w_hi = m0.i + t0;    // window (all-ones Hann of length 1, see fx.py)
    // sweep 2: energies and peaks
    float a_sum = 0.f, c_sum = 0.f, late_sq = 0.f, mx_late = 0.f, mx_ew = 0.f, mx_enw = 0.f;
    auto acc_one = [&](float v, int64_t i) {
      const bool early = i >= e_lo && i <= e_hi;
      const bool win = i >= w_lo && i <= w_hi;
      const float av = fabsf(v), sq = v * v;
      // branch-free: every element lands in exactly one class
      a_sum += (early && win) ? sq : 0.f;
      c_sum += (early && !win) ? sq : 0.f;
      late_sq += early ? 0.f : sq;
      mx_ew = fmaxf(mx_ew, (early && win) ? av : 0.f);
      mx_enw = fmaxf(mx_enw, (early && !win) ? av : 0.f);
      mx_late = fmaxf(mx_late, early ? 0.f : av);
    };
    if (vec) {
      const float4* __restrict__ p4 = reinterpret_cast<const float4*>(xr);
      const int64_t T4 = T / 4;
      int64_t i = threadIdx.x;
      for (; i + RT < T4; i += 2 * RT) {
        const float4 a = p4[i], b = p4[i + RT];
        acc_one(a.x, 4 * i); acc_one(a.y, 4 * i + 1); acc_one(a.z, 4 * i + 2); acc_one(a.w, 4 * i + 3);
        const int64_t j = i + RT;
        acc_one(b.x, 4 * j); acc_one(b.y, 4 * j + 1); acc_one(b.z, 4 * j + 2); acc_one(b.w, 4 * j + 3);
      }
      for (; i < T4; i += RT) {
        const float4 a = p4[i];
        acc_one(a.x, 4 * i); acc_one(a.y, 4 * i + 1); acc_one(a.z, 4 * i + 2); acc_one(a.w, 4 * i + 3);
      }
    } else {
      for (int64_t i = threadIdx.x; i < T; i += RT) acc_one(xr[i], i);
    }
    a_sum = block_sum(a_sum, shf);
    c_sum = block_sum(c_sum, shf);
    late_sq = block_sum(late_sq, shf);
    mx_late = block_max(mx_late, shf);
    mx_ew = block_max(mx_ew, shf);
    mx_enw = block_max(mx_enw, shf);
    // alpha: larger root of a al^2 + b al + c = 0 with b = 0 (the window is 0/1), floored by
    // max|late| / max|early|  (effects.py:601-607, 636-640).  NaNs propagate like torch.maximum.
    const float cc = c_sum - powf(10.f, drr[b] / 10.f) * late_sq;
    const float disc = sqrtf(0.f * 0.f - 4.f * a_sum * cc);
    const float r1 = (-0.f - disc) / (2.f * a_sum), r2 = (-0.f + disc) / (2.f * a_sum);
    float alpha = (r1 != r1 || r2 != r2) ? NAN : fmaxf(r1, r2);
    const float min_alpha = mx_late / fmaxf(mx_ew, mx_enw);
    alpha = (alpha != alpha || min_alpha != min_alpha) ? NAN : fmaxf(alpha, min_alpha);
    // ensure_max_of_audio(1.0): peak of the result, analytically (multiplication is monotonic)
    const float peak = fmaxf(fmaxf(fabsf(alpha * mx_ew), mx_enw), mx_late);
    const float gain = (alpha == alpha && peak > 1.0f) ? 1.0f / peak : 1.0f;  // NaN peak: no rescale
    // sweep 3: literal alpha*w*early + (1-w)*early + late (a non-finite alpha poisons the row as in torch)
    float* __restrict__ orow = out + row * T;
    // max |output| and its first position (absmax_kernel's rule), for the convolution that follows in apply_ir: the
    // fourth pass over the impulse responses (0.11 ms at cfg4) rides on this one
    MaxIdx am{-1.0f, 0};
    auto out_one = [&](float v, int64_t i) {
      const bool early = i >= e_lo && i <= e_hi;
      const bool win = i >= w_lo && i <= w_hi;
      const float e = early ? v : 0.f, l = early ? 0.f : v, w = win ? 1.f : 0.f;
      const float o = (alpha * w * e + (1.f - w) * e + l) * gain;
      const float a = fabsf(o);
      if (takes_over(a, am.v)) { am.v = a; am.i = i; }
      return o;
    };
    if (vec) {
      const float4* __restrict__ p4 = reinterpret_cast<const float4*>(xr);
      float4* __restrict__ o4 = reinterpret_cast<float4*>(orow);
      const int64_t T4 = T / 4;
      int64_t i = threadIdx.x;
      for (; i + RT < T4; i += 2 * RT) {
        const float4 a = p4[i], b = p4[i + RT];
        const int64_t j = i + RT;
        o4[i] = make_float4(out_one(a.x, 4 * i), out_one(a.y, 4 * i + 1), out_one(a.z, 4 * i + 2), out_one(a.w, 4 * i + 3));
        o4[j] = make_float4(out_one(b.x, 4 * j), out_one(b.y, 4 * j + 1), out_one(b.z, 4 * j + 2), out_one(b.w, 4 * j + 3));
      }
      for (; i < T4; i += RT) {
        const float4 a = p4[i];
        o4[i] = make_float4(out_one(a.x, 4 * i), out_one(a.y, 4 * i + 1), out_one(a.z, 4 * i + 2), out_one(a.w, 4 * i + 3));
      }
    } else {
      for (int64_t i = threadIdx.x; i < T; i += RT) orow[i] = out_one(xr[i], i);
    }
    if (vmax) {
      am = block_argmax(am, shm);
      if (threadIdx.x == 0) {
        vmax[row] = am.v;
        if (imax) imax[row] = am.i;
      }
    }
  }
}

// ---- windows along the batch axis (dsp.py:70-151) --------------------------------------------------
// collect: out[(r nw + w), n] = x[r, w hop + n]  (torch unfold + permute + reshape: one gather).
__global__ __launch_bounds__(256) void collect_windows_kernel(const float* __restrict__ x, int64_t T, int win, int hop, int64_t nw,
                                                              float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t fr = i / win;
    const int n = (int)(i - fr * win);
    const int64_t r = fr / nw, w = fr - r * nw;
    out[i] = x[r * T + w * hop + n];
  }
}

// overlap-add: the reference folds the windows, folds a tensor of ones, divides, then trims `trim` samples from either
// end -- four passes over window-sized data.  Here every output sample gathers its <= ceil(win / hop) windows (ascending
// window order) and divides by their number; positions no window covers are 0 / 0 = NaN, as in the reference.
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ fr, int64_t nw, int win, int hop, int64_t trim,
                                                          int64_t out_len, float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / out_len;
    const int64_t s = i - r * out_len + trim;          // position in the padded signal
    int64_t w_hi = s / hop;
    if (w_hi > nw - 1) w_hi = nw - 1;
    int64_t w_lo = (s - win + hop) / hop;              // smallest w with w hop + win > s
    if (s - win + 1 <= 0 || w_lo < 0) w_lo = 0;
    float acc = 0.f, cnt = 0.f;
    const float* __restrict__ base = fr + r * nw * win;
    for (int64_t w = w_lo; w <= w_hi; ++w) {
      const int64_t n = s - w * hop;
      if (n >= 0 && n < win) { acc += base[w * win + n]; cnt += 1.f; }
    }
    out[i] = acc / cnt;
  }
}

// ---- quantization / mu-law quantization (effects.py:452-527) -------------------------------------
// The reference's chains of ~10 / ~22 whole-tensor torch operations (one of them through int64), one float32 operation
// per step and in the same order, so the values equal the chain's; out = a - (a - q(a)) is its straight-through form.
__device__ __forceinline__ float sgn_torch(float v) { return (float)((0.f < v) - (v < 0.f)); }   // torch.sign: 0 for NaN

template <int MODE>
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ x, const float* __restrict__ q, int64_t per_item,
                                                       float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float a = x[i];
    const float qb = q[i / per_item];
    float v;
    if (MODE == 0) {
      v = floorf((a + 1.f) / 2.f * qb) / qb;
      v = 2.f * v - 1.f;
    } else {
      const float mu = qb, l1 = log1pf(mu);
      v = sgn_torch(a) * log1pf(mu * fabsf(a)) / l1;
      const long long lv = (long long)__fadd_rn(__fmul_rn((v + 1.f) / 2.f, mu), 0.5f);    // (no fused multiply-add: the chain rounds twice)
      v = ((float)lv / mu) * 2.f - 1.f;
      v = sgn_torch(v) * (expf(fabsf(v) * l1) - 1.f) / mu;
    }
    const float r = a - v;
    out[i] = a - r;
  }
}

// The same with the ROW IN REGISTERS (round 5): mono impulse responses of up to RT * NV float4 -- a 2 s RIR at 48 kHz is
// 96 000 samples = 47 float4 per thread -- are read ONCE; the arg-max, the energies and the output all come from the
// registers (the three-sweep kernel above moved 1.61 GB for the 0.79 GB of one read + one write at cfg4: its second and
// third sweeps missed L2).  Every thread holds exactly the elements the sweeps of alter_drr_kernel hand it, in the same
// order, and the block reductions are the same functions: the results are bit-identical.
template <int NV>
__global__ __launch_bounds__(RT, 2) void alter_drr_regs_kernel(const float* __restrict__ x, int64_t B, int64_t T, int t0,
                                                               const float* __restrict__ drr /* (B) */, float* __restrict__ out,
                                                               float* __restrict__ vmax, int64_t* __restrict__ imax) {
  __shared__ MaxIdx shm[RT / 64];
  __shared__ float shf[RT / 64];
  const int T4 = (int)(T / 4);
  const int tid = (int)threadIdx.x;
  const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);      // first float4 slot of this wave in every k-iteration
  // k-iterations: [0, kfull) hold a float4 for every lane of this wave, iteration kfull (if < per) only for some lanes,
  // everything behind it for none -- all three are wave-uniform, so the sweeps below carry lane masks in ONE iteration
  const int dfull = T4 - wbase - 64;
  const int kfull = dfull >= 0 ? min(NV, dfull / RT + 1) : 0;
  const int kpart = (kfull < NV && wbase + RT * kfull < T4) ? 1 : 0;
  for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(x + row * T);
    float4 r[NV];
    int t0_ = tid;                                       // opaque per sweep: the per-element indices are recomputed here, not
    asm volatile("" : "+v"(t0_));                        // hoisted out of the row loop and kept alive next to the row itself
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = t0_ + RT * k;
      r[k] = p4[i < T4 ? i : T4 - 1];                    // clamped, unconditional: all loads in flight at once
    }
    // sweep 1 (registers): signed arg-max, 32-bit indices (T < 2^31)
    float bv = -INFINITY;
    int bi = 0;
    int t1 = tid;
    asm volatile("" : "+v"(t1));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < kfull + kpart) {                           // wave-uniform
        const int i = t1 + RT * k;
        const bool ok = k < kfull || i < T4;
        const float e[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool take = ok && e[u] > bv;
          bv = take ? e[u] : bv;
          bi = take ? 4 * i + u : bi;
        }
      }
    }
    const MaxIdx m = block_argmax(MaxIdx{bv, (int64_t)bi}, shm);
    const int e_lo = (int)m.i - t0, e_hi = (int)m.i + t0;           // early span = window (mono: channel 0 is this row)
    const unsigned e_span = 2u * (unsigned)t0;                      // early  <=>  (unsigned)(n - e_lo) <= e_span
    // does iteration k of THIS wave touch the early span at all?  (wave-uniform; true for one or two iterations of one or
    // two waves: everywhere else an element is "late" and the sweeps take their three-instruction path)
    auto touches = [&](int k) { const int lo = 4 * (wbase + RT * k); return lo <= e_hi && lo + 255 >= e_lo; };
    // sweep 2 (registers): energies and peaks
    float a_sum = 0.f, late_sq = 0.f, mx_late = 0.f, mx_ew = 0.f;
    int t2 = tid;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < kfull + kpart) {
        const int i = t2 + RT * k;
        const bool ok = k < kfull || i < T4;
        const float e[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
        if (!touches(k)) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float v = ok ? e[u] : 0.f;             // (a surplus element adds +0 to a sum and max(., 0) to a peak: neutral)
            late_sq += v * v;
            mx_late = fmaxf(mx_late, fabsf(v));
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool early = (unsigned)(4 * i + u - e_lo) <= e_span;
            const float v = ok ? e[u] : 0.f;
            const float av = fabsf(v), sq = v * v;
            a_sum += early ? sq : 0.f;
            late_sq += early ? 0.f : sq;
            mx_ew = fmaxf(mx_ew, early ? av : 0.f);
            mx_late = fmaxf(mx_late, early ? 0.f : av);
          }
        }
      }
    }
    a_sum = block_sum(a_sum, shf);
    const float c_sum = 0.f;                              // early-but-outside-the-window is empty for a mono row
    late_sq = block_sum(late_sq, shf);
    mx_late = block_max(mx_late, shf);
    mx_ew = block_max(mx_ew, shf);
    const float mx_enw = 0.f;
    const float cc = c_sum - powf(10.f, drr[row] / 10.f) * late_sq;
    const float disc = sqrtf(0.f * 0.f - 4.f * a_sum * cc);
    const float r1 = (-0.f - disc) / (2.f * a_sum), r2 = (-0.f + disc) / (2.f * a_sum);
    float alpha = (r1 != r1 || r2 != r2) ? NAN : fmaxf(r1, r2);
    const float min_alpha = mx_late / fmaxf(mx_ew, mx_enw);
    alpha = (alpha != alpha || min_alpha != min_alpha) ? NAN : fmaxf(alpha, min_alpha);
    const float peak = fmaxf(fmaxf(fabsf(alpha * mx_ew), mx_enw), mx_late);
    const float gain = (alpha == alpha && peak > 1.0f) ? 1.0f / peak : 1.0f;
    // what alpha * w * e + (1 - w) * e contributes to a late element (w = 0, e = 0): zero, or NaN when alpha is not finite
    // (a non-finite alpha poisons the whole row, as in torch)
    const float late_term = alpha * 0.f * 0.f + 1.f * 0.f;
    // sweep 3 (registers -> out)
    float4* __restrict__ o4 = reinterpret_cast<float4*>(out + row * T);
    float av_best = -1.0f;
    int ai = 0;
    int t3 = tid;
    asm volatile("" : "+v"(t3));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < kfull + kpart) {
        const int i = t3 + RT * k;
        const bool ok = k < kfull || i < T4;
        const float e[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
        float o[4];
        if (!touches(k)) {
#pragma unroll
          for (int u = 0; u < 4; ++u) o[u] = (late_term + e[u]) * gain;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool early = (unsigned)(4 * i + u - e_lo) <= e_span;
            const float ev = early ? e[u] : 0.f, l = early ? 0.f : e[u], w = early ? 1.f : 0.f;
            o[u] = (alpha * w * ev + (1.f - w) * ev + l) * gain;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float a = fabsf(o[u]);
          const bool take = ok && takes_over(a, av_best);
          av_best = take ? a : av_best;
          ai = take ? 4 * i + u : ai;
        }
        if (ok) o4[i] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (vmax) {
      const MaxIdx am = block_argmax(MaxIdx{av_best, (int64_t)ai}, shm);
      if (tid == 0) {
        vmax[row] = am.v;
        if (imax) imax[row] = am.i;
      }
    }
  }
}

}  // namespace

extern "C" {

// x (rows, T) -> vmax[rows] = max_n |x[r, n]|, imax[rows] = first n attaining it (imax may be NULL)
int at_absmax_f32(const float* x, int64_t rows, int64_t T, float* vmax, int64_t* imax, void* stream) {
  if (rows == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !vmax || rows < 0 || T <= 0) return AT_ERR_INVALID;
  if (rows == 0) return AT_OK;
  const int64_t blocks = rows < 65536 ? rows : 65536;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(RT), 0, reinterpret_cast<hipStream_t>(stream), x, rows,
                     T, vmax, imax);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// x (rows, L) -> out (rows, T): the row zero-padded / truncated to T, then rotated left by
// shift[r] (NULL = no rotation):  out[r, n] = xz[r, (n + shift[r]) mod T]
int at_roll_pad_f32(const float* x, int64_t rows, int64_t L, const int64_t* shift, int64_t T, float* out, void* stream) {
  if (rows == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !out || rows < 0 || L <= 0 || T <= 0 || rows > 65535LL * 1024) return AT_ERR_INVALID;
  if (rows == 0) return AT_OK;
  if (rows > 65535) return AT_ERR_UNSUPPORTED;
  int64_t bx = (T + 256 * 8 - 1) / (256 * 8);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(roll_pad_kernel, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, rows, L, shift, T, out);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

// x (B, C, T) impulse responses, drr (B) target direct-to-reverberant ratios in dB, t0 = early
// half-span in samples (int(sample_rate * 0.0025)).  out (B, C, T) = alter_drr(x) followed by
// ensure_max_of_audio(1.0).  out may alias x.
// vmax (B*C) / imax (B*C), both optional: max |out| per row and its first position, as at_absmax_f32(out) would report.
int at_alter_drr_peak_f32(const float* x, int64_t B, int64_t C, int64_t T, int t0, const float* drr, float* out,
                          float* vmax, int64_t* imax, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !drr || !out || B < 0 || C <= 0 || T <= 0 || t0 < 0 || (imax && !vmax)) return AT_ERR_INVALID;
  if (C > 1 && x == out) return AT_ERR_INVALID;  // channel 0 is re-read by the other channels
  const int64_t rows = B * C;
  const int64_t blocks = rows < 65536 ? rows : 65536;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // mono rows that fit the register file of one workgroup: one read, one write
  const bool aligned = (T % 4) == 0 && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
  if (C == 1 && aligned && T / 4 <= (int64_t)RT * 48 && T < (1LL << 31)) {
    const int64_t per = (T / 4 + RT - 1) / RT;
    if (per <= 16) hipLaunchKernelGGL(alter_drr_regs_kernel<16>, dim3((unsigned)blocks), dim3(RT), 0, st, x, B, T, t0, drr, out, vmax, imax);
    else if (per <= 32) hipLaunchKernelGGL(alter_drr_regs_kernel<32>, dim3((unsigned)blocks), dim3(RT), 0, st, x, B, T, t0, drr, out, vmax, imax);
    else hipLaunchKernelGGL(alter_drr_regs_kernel<48>, dim3((unsigned)blocks), dim3(RT), 0, st, x, B, T, t0, drr, out, vmax, imax);
    AT_LAUNCH_CHECK();
    return AT_OK;
  }
  hipLaunchKernelGGL(alter_drr_kernel, dim3((unsigned)blocks), dim3(RT), 0, st, x, B,
                     (int)C, T, t0, drr, out, vmax, imax);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

int at_alter_drr_f32(const float* x, int64_t B, int64_t C, int64_t T, int t0, const float* drr, float* out,
                     void* stream) {
  return at_alter_drr_peak_f32(x, B, C, T, t0, drr, out, nullptr, nullptr, stream);
}

int at_collect_windows_f32(const float* x, int64_t rows, int64_t T, int win, int hop, float* out, void* stream) {
  if (rows == 0) return AT_OK;
  if (!x || !out || rows < 0 || T <= 0 || win <= 0 || hop <= 0) return AT_ERR_INVALID;
  if (T < win) return AT_OK;                           // no window fits: the result is empty
  const int64_t nw = (T - win) / hop + 1;
  const int64_t total = rows * nw * win;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(collect_windows_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, T, win,
                     hop, nw, out, total);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

int at_overlap_add_f32(const float* frames, int64_t rows, int64_t nw, int win, int hop, int64_t trim, int64_t out_len, float* out,
                       void* stream) {
  if (rows == 0 || out_len == 0) return AT_OK;
  if (!frames || !out || rows < 0 || nw <= 0 || win <= 0 || hop <= 0 || trim < 0 || out_len < 0) return AT_ERR_INVALID;
  const int64_t total = rows * out_len;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), frames, nw,
                     win, hop, trim, out_len, out, total);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

int at_quantize_f32(const float* x, int64_t B, int64_t per_item, const float* q, int mulaw, float* out, void* stream) {
  if (B == 0 || per_item == 0) return AT_OK;
  if (!x || !q || !out || B < 0 || per_item < 0) return AT_ERR_INVALID;
  const int64_t total = B * per_item;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (mulaw) hipLaunchKernelGGL(quantize_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, q, per_item, out, total);
  else hipLaunchKernelGGL(quantize_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, x, q, per_item, out, total);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

}  // extern "C"
