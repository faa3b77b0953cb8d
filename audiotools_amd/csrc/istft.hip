// Inverse STFT for gfx950 (power-of-two n_fft in [32, 2048]).
//
// Replaces reference audiotools/core/audio_signal.py:1283-1290  torch.istft(X, n_fft, hop,
// window, length, center=True): per frame a C2R FFT, multiplication by the window, overlap-add,
// division by the overlap-added squared window, removal of the n_fft/2 centre padding.
//
// Two code paths:
//  * hop = n_fft / {2,4,8,16}: FUSED kernel (istft_fused_kernel).  A frame slot (L lanes) walks a
//    run of consecutive frames of one row; the windowed frames are overlap-added in a register
//    window of n_fft samples that shifts by one hop per frame (the mirror image of the register
//    reuse in the forward kernel), finished hops are scaled by the reciprocal envelope and stored.
//    HBM sees X once (+ (n_fft/hop - 1)/run of re-read at run boundaries) and the signal once:
//    no frame buffer, no second pass.
//  * any other hop: kernel A + kernel B below through a (rows, frames, n_fft) frame buffer.
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
#include "generic_fft.h"
#include <type_traits>

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

// ---- fused inverse: FFT + window + overlap-add in registers ---------------------------------
struct IstftFusedArgs {
  const float2* X;        // (rows, n_x, M+1) bin-contiguous
  const float* window;    // (N)
  const float2* tw;       // (N)
  const float* inv_env;   // ((n_frames-1)*hop + N): 1 / sum_f w^2, 0 where the envelope vanishes
  float* dump;            // 1024 floats behind the envelope table: per-lane, per-q sink of edge-step pair stores (never read)
  const float2* zeros;    // ZERO_PAGE_FLOATS zero floats behind the dump slots (written by istft_env_kernel): the "spectrum"
                          // of the virtual all-zero frames
  float* out;             // (rows, length)
  int64_t rows;
  int64_t length;
  int n_x;                // frames stored in X
  int lead;               // virtual all-zero frames in front of X's first frame (match_stride: 2)
  int n_frames;           // virtual frames in total (lead + n_x + trailing zero frames)
  int n_seg;              // hop-sized output segments per row = n_frames - 1 + N/hop
  int run;                // segments per unit
  int runs_per_row;
  int64_t total_units;    // rows * runs_per_row
  // mel backward (MELB): X holds the SAVED forward spectrum, the gradient comes from the mel output
  const float* gmel;      // (rows, n_x, n_mels) dL/dmel, band-contiguous
  const int* bin_bands;   // (M+1): the two bands whose triangles cover bin k: lo | hi << 16
  const float2* bin_w;    // (M+1): their weights basis[lo][k], basis[hi][k] (0 where absent)
  int n_mels;
  // STFT-domain edit applied to the spectrum as it is loaded (EDIT kernels): SpectralTransform's
  // stft -> edit -> istft (transforms.py:274-286) without the edit's own pass over stft_data (dsp.py:217-352)
  int edit_kind;           // 1: bins [lo, hi) of every frame := fill;  2: frames [lo, hi) := fill;
                           // 3: X *= e^{i shift};  4: |X| in dB below the cutoff -> val e^{i angle X}
  int edit_C;              // channels per item: the parameters are per ITEM, row / C
  const int* edit_lo;      // (B) kinds 1, 2
  const int* edit_hi;
  const float* edit_shift; // (B) kind 3
  const double* edit_cut;  // (B) kind 4: cutoff in dB
  const unsigned* edit_maxpow;   // kind 4: max |X|^2 over the whole batch as float bits (at_spec_maxpow_f32)
  float2 edit_fill;
  float edit_top_db, edit_val;
  int edit_use_top;
};

struct __attribute__((packed, aligned(4))) f2u { float x, y; };  // dword-aligned pair

// envelope: every padded position gathers w^2 of the <= N/hop frames that cover it
constexpr int DUMP_FLOATS = 1024;
constexpr int ZERO_PAGE_FLOATS = 2304;   // >= 2 (1024 + 1): one frame of the largest fused size

__global__ __launch_bounds__(256) void istft_env_kernel(const float* __restrict__ window, float* __restrict__ inv_env,
                                                        int n_frames, int N, int hop, int64_t total) {
  // the zero page behind the envelope table and the dump slots
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ZERO_PAGE_FLOATS; i += (int64_t)gridDim.x * blockDim.x)
    inv_env[total + DUMP_FLOATS + i] = 0.f;
  for (int64_t pp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pp < total; pp += (int64_t)gridDim.x * blockDim.x) {
    int64_t f_hi = pp / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    int64_t f_lo = (pp - N + hop) / hop;
    if (pp - N + 1 <= 0) f_lo = 0;
    if (f_lo < 0) f_lo = 0;
    float env = 0.f;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
      const int n = (int)(pp - f * hop);
      if (n < 0 || n >= N) continue;
      const float w = window[n];
      env = fmaf(w, w, env);
    }
    inv_env[pp] = env > 1e-11f ? 1.0f / env : 0.f;
  }
}

// cache policy of the streaming traffic of the fused kernel: non-temporal spectrum loads and signal
// stores measured 2.097 -> 2.056 ms at B = 512 (same box, profiles/r02_notes.md); 0 = plain
#ifndef AT_NT_ISTFT_LD
#define AT_NT_ISTFT_LD 1
#endif
#ifndef AT_NT_ISTFT_ST
#define AT_NT_ISTFT_ST 1
#endif
#ifndef AT_ISTFT_WPS
#define AT_ISTFT_WPS 2
#endif

// ADJ = true turns the kernel into the ADJOINT of the forward transform (backward pass of stft()):
// x_bar = OLA(window * sum_k Re(G_k e^{+2 pi i k n / N})) -- same data flow, but no 1/N, no factor 2 on
// the interior bins (i.e. DC and Nyquist count double relative to c2r), no envelope division and no
// removal of the centre padding (out covers the whole padded signal).
// MELB (with ADJ): backward of the FUSED mel path.  X is the saved forward spectrum and the incoming
// gradient is dL/dmel; the spectrum gradient  G[k] = (sum_m basis[m,k] gmel[m]) X[k] / |X[k]|  is formed
// on the fly from a per-bin table (a triangular bank has at most two non-zero bands per bin) and
// never touches HBM.
template <int M, int SH /* hop = 2 L SH */, bool ADJ, bool MELB = false, bool EDIT = false>
__global__ __launch_bounds__(256, AT_ISTFT_WPS) void istft_fused_kernel(const IstftFusedArgs A) {
  using P = Plan<M>;
  constexpr int L = P::L, FW = P::FW, N = 2 * M, HOP = 2 * L * SH, R = N / HOP;
  constexpr bool DUMP = !ADJ;   // the inverse transform has a workspace (envelope table + dump slots behind it)
  // A virtual all-zero frame (match_stride's edge frames, the lead-in / lead-out steps of a run) LOADS zeros from a
  // zero page of the workspace instead of loading frame 0 and being zeroed afterwards.  At M = 1024 a wave holds one
  // frame, "dead" is wave-uniform, and the compiler turned the zeroing into 16 uniform branches per frame -- one
  // around every fold butterfly --, i.e. 16 basic blocks whose two ds_bpermute + one ds_read each had to come back
  // before the next could be issued.  Without the zeroing the fold is one block: 32 + 16 LDS operations in flight
  // (1.963 -> 1.954 ms at B = 512, same box interleaved, profiles/r03_notes.md).
  constexpr bool ZERO_PAGE = !ADJ && !MELB && !EDIT;
  // PAIRED (M = 16 L = 1024, 512): the transform runs as 4 . 16 . (M / 64) instead of 16 . 16 . (M / 256), and a lane owns
  // the four radix-4 groups
  //   j = t, t + L, 3L - t, 4L - t   (lane 0: 0, L, 3L, 2L),   points j + 4L r in register b + 4 r,
  // a set closed under k -> M - k: the Hermitian partner of register q is register 15 - q of the SAME lane (lane 0:
  // groups 0 and 2L pair with themselves, bin 0 with the Nyquist bin).  The fold needs no lane exchange -- the 32
  // ds_bpermute per frame of the t + 64 q layout are gone --, the pairs of the two middle groups come out of one
  // evaluation each, and all four groups are still 512-byte runs of a frame's bins (two ascending, two descending).
  // The first pass writes every group where the Stockham order wants it, so nothing behind it changes but the radices.
  // (M = 512: radices 4 . 16 . 8; the last pass composes its twiddles from three table entries per butterfly -- with all 14
  //  read up front the hop = n_fft / 4 instantiation needs 258 registers and spills the row pointer, reloaded in front of every
  //  prefetch behind an s_waitcnt vmcnt(0).  Its EDIT instantiation keeps the exchange layout: no GPU test covers folded edits at
  //  n_fft 1024.  M = 256 would be 4 . 16 . 4: tools/emulate_istft_paired.py replays all three.)
  constexpr bool PAIRED = (M == 1024 || (M == 512 && !EDIT)) && !ADJ && !MELB;
  constexpr int R3P = PAIRED ? M / 64 : 16;                       // PAIRED: the last radix
  __shared__ float2 lds[4 * WAVE_LDS_SLOTS];
  __shared__ float2 s_win2[M];
  __shared__ float2 s_twf[M];                                     // fold twiddles (cos, -sin)(2 pi k / N), k < M
  __shared__ __attribute__((aligned(16))) float s_tw2[16 * 36];   // pass-2 twiddles, row = j mod 16 (PAIRED: j mod 4)
  __shared__ __attribute__((aligned(16))) float s_tw3[PAIRED ? L * 36 : 4];    // PAIRED: last-pass twiddles, row = t
  __shared__ int s_bb[MELB ? M + 1 : 1];
  __shared__ float2 s_bw[MELB ? M + 1 : 1];
  if constexpr (MELB) {
    for (int i = threadIdx.x; i <= M; i += 256) { s_bb[i] = A.bin_bands[i]; s_bw[i] = A.bin_w[i]; }
  }
  // the 1/N of the inverse transform (1/2 for the adjoint) rides on the window table
  const float inv_n = ADJ ? 0.5f : 1.0f / (float)(2 * M);
  for (int i = threadIdx.x; i < M; i += 256) {
    const float2 w = reinterpret_cast<const float2*>(A.window)[i];
    s_win2[i] = make_float2(w.x * inv_n, w.y * inv_n);
    s_twf[i] = A.tw[i];
  }
  if constexpr (PAIRED) {
    for (int i = threadIdx.x; i < 4 * 16; i += 256) {             // second pass: radix 16 behind NS = 4
      const int jj = i / 16, r = i % 16;
      reinterpret_cast<float2*>(s_tw2 + jj * 36)[r] = A.tw[r * jj * (N / 64)];
    }
    for (int i = threadIdx.x; i < L * 16; i += 256) {             // last pass: radix M / 64 behind NS = 64, entry b R + r
      const int tt = i / 16, b3 = (i % 16) / R3P, r = i % R3P;
      reinterpret_cast<float2*>(s_tw3 + tt * 36)[i % 16] = A.tw[r * ((tt + b3 * L) % 64) * (N / (64 * R3P))];
    }
  } else if constexpr (P::R2 > 1) {
    for (int i = threadIdx.x; i < 16 * P::R2; i += 256) {
      const int jj = i / P::R2, r = i % P::R2;
      reinterpret_cast<float2*>(s_tw2 + jj * 36)[r] = A.tw[r * jj * (N / (16 * P::R2))];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fs = lane / L, t = lane % L;
  float2* fbuf = lds + wave * WAVE_LDS_SLOTS + fs * P::SLOTS;
  constexpr int NB2 = 16 / P::R2;
  constexpr int NB3 = 16 / P::R3;

  // this frame slot's unit: (row, run of segments [h0, h1))
  int64_t unit = ((int64_t)blockIdx.x * 4 + wave) * FW + fs;
  const bool unit_ok = unit < A.total_units;
  if (!unit_ok) unit = A.total_units - 1;   // keep the lanes converged; nothing is stored
  const int64_t row = unit / A.runs_per_row;
  const int h0 = (int)(unit - row * A.runs_per_row) * A.run;
  const int h1 = min(h0 + A.run, A.n_seg);
  const float2* __restrict__ Xrow = A.X + row * (int64_t)A.n_x * (M + 1);
  float* __restrict__ orow = A.out + row * A.length;
  const int len = (int)A.length;
  // edit parameters of this frame slot's item (constant over the run)
  int e_lo = 0, e_hi = 0;
  float e_cs = 1.f, e_sn = 0.f, e_floor = 0.f;
  double e_cut = 0.0;
  if constexpr (EDIT) {
    const int64_t item = row / A.edit_C;
    if (A.edit_kind == 1 || A.edit_kind == 2) { e_lo = A.edit_lo[item]; e_hi = A.edit_hi[item]; }
    if (A.edit_kind == 3) sincosf(A.edit_shift[item], &e_sn, &e_cs);
    if (A.edit_kind == 4) {
      e_cut = A.edit_cut[item];
      e_floor = 10.0f * log10f(fmaxf(__uint_as_float(*A.edit_maxpow), 1e-10f)) - A.edit_top_db;
    }
  }

  float2 tw3b[P::R3 > 1 ? NB3 : 1];
  if constexpr (P::R3 > 1) {
    constexpr int NS = 16 * P::R2;
#pragma unroll
    for (int b = 0; b < NB3; ++b) tw3b[b] = A.tw[((t + b * L) % NS) * (N / (NS * P::R3))];
  }

  float2 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = make_float2(0.f, 0.f);

  // Every lane loads bin k = t + L q of its frame ONCE (ascending, 512 B per wave instruction);
  // the Hermitian partner X[M-k] of the fold is register 15-q of lane (L - t) mod L and comes
  // over the LDS crossbar (ds_bpermute).  Lane 0 pairs with itself: register 16-q, and the
  // Nyquist bin for q = 0.  Loads are branch-free (a dead frame reads frame 0 and is zeroed by
  // select) and issued one frame AHEAD, right after the fold consumed the registers, so their
  // latency hides behind the FFT of the current frame.
  const int src_lane = (lane - t) + ((L - t) & (L - 1));
  // bin held by register q: t + L q, or (PAIRED) group b = q & 3, r = q >> 2
  const int pj2 = 3 * L - t, pj3 = t == 0 ? 2 * L : 4 * L - t;
  auto kq = [&](int q) __attribute__((always_inline)) -> int {
    if constexpr (!PAIRED) return t + L * q;
    const int b = q & 3, r = q >> 2;
    return (b == 0 ? t : b == 1 ? t + L : b == 2 ? pj2 : pj3) + 4 * L * r;
  };
  float2 xa[16], xN;
  constexpr int GP = 8;            // mel values per lane: n_mels <= GP * L
  float gpre[MELB ? GP : 1];
  auto issue_loads = [&](int f) __attribute__((always_inline)) -> bool {
    const int fx = f - A.lead;
    const bool live = fx >= 0 && fx < A.n_x && f < h1;
    const float2* __restrict__ Xf = (ZERO_PAGE && !live) ? A.zeros : Xrow + (int64_t)(live ? fx : 0) * (M + 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) xa[q] = at::ldg2<AT_NT_ISTFT_LD != 0>(Xf + kq(q));
    xN = at::ldg2<AT_NT_ISTFT_LD != 0>(Xf + M);
    if constexpr (MELB) {
      const float* __restrict__ gr = A.gmel + (row * (int64_t)A.n_x + (live ? fx : 0)) * A.n_mels;
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int m = t + L * j;
        const float v = gr[m < A.n_mels ? m : 0];    // branch-free: clamped address, select
        gpre[j] = m < A.n_mels ? v : 0.f;
      }
    }
    return live;
  };
  bool live_nxt = issue_loads(h0 - (R - 1));
  // One step = one virtual frame added = one segment finished.  Two things in it are shaped by the memory
  // pipeline rather than by the arithmetic (vmcnt retires IN ORDER, and the compiler can only count memory
  // operations that are issued on every path; profiles/r03_notes.md):
  //  * the envelope loads are issued BEFORE the spectrum prefetch of the next frame: the stores of this step
  //    need them, and behind the prefetch they forced all 17 prefetch loads to complete before the first store;
  //  * every path issues exactly SH pair stores (edge steps redirect pairs outside the row to a dump slot of
  //    the workspace instead of branching around them) and the first step is peeled, so both edges of the loop
  //    carry "prefetch loads, then SH stores": the loop head waits with vmcnt(SH + ...) for the loads only.
  //    With guarded stores the head was s_waitcnt vmcnt(0): the previous frame's stores drained every frame.
  auto step = [&](int f) __attribute__((always_inline)) {
    const bool live = live_nxt;
    if constexpr (MELB) {
      // dL/dmel row of this frame -> the frame slot's slab (free between frames), then per bin
      // g_mag = w_lo g[lo] + w_hi g[hi]  and  G = g_mag X / |X|  (0 where X == 0, as torch's abs)
      float* gs = reinterpret_cast<float*>(fbuf);
#pragma unroll
      for (int j = 0; j < GP; ++j)
        if (t + L * j < A.n_mels) gs[t + L * j] = gpre[j];
      wave_sync();
      auto to_grad = [&](float2 x, int k) {
        const int bb = s_bb[k];
        const float2 bw = s_bw[k];
        const float gm = fmaf(bw.x, gs[bb & 0xffff], bw.y * gs[bb >> 16]);
        const float p = fmaf(x.x, x.x, x.y * x.y);
        const float sc = p > 0.f ? gm * __builtin_amdgcn_rsqf(p) : 0.f;
        return make_float2(x.x * sc, x.y * sc);
      };
#pragma unroll
      for (int q = 0; q < 16; ++q) xa[q] = to_grad(xa[q], t + L * q);
      xN = to_grad(xN, M);
      wave_sync();   // the slab is about to be overwritten by pass 1
    }
    if constexpr (EDIT) {
      // the arithmetic of spec_mask_kernel / spec_phase_shift_kernel / spec_mask_lowmag_kernel (specedit.hip) on the
      // registers the loads filled; masked / unmasked elements are the same ones (integer ranges = the reference's
      // float comparisons against a monotone grid, evaluated by the caller)
      const int fx = f - A.lead;
      auto edit = [&](float2 x, int k) __attribute__((always_inline)) -> float2 {
        if (A.edit_kind == 1) return (k >= e_lo && k < e_hi) ? A.edit_fill : x;
        if (A.edit_kind == 2) return (fx >= e_lo && fx < e_hi) ? A.edit_fill : x;
        if (A.edit_kind == 3) return make_float2(x.x * e_cs - x.y * e_sn, x.x * e_sn + x.y * e_cs);
        const float mag = hypotf(x.x, x.y);
        float ls = 10.0f * log10f(fmaxf(mag * mag, 1e-10f));
        if (A.edit_use_top) ls = fmaxf(ls, e_floor);
        if ((double)ls < e_cut) {
          const float inv = mag > 0.f ? 1.0f / mag : 0.f;
          return mag > 0.f ? make_float2(A.edit_val * x.x * inv, A.edit_val * x.y * inv) : make_float2(A.edit_val, 0.f);
        }
        return x;
      };
#pragma unroll
      for (int q = 0; q < 16; ++q) xa[q] = edit(xa[q], kq(q));
      xN = edit(xN, M);
    }
    float2 a[16];
    if constexpr (PAIRED) {
      if constexpr (!ZERO_PAGE) {
        if (!live) {
#pragma unroll
          for (int q = 0; q < 16; ++q) xa[q] = make_float2(0.f, 0.f);
          xN = make_float2(0.f, 0.f);
        }
      }
      // the two middle groups: registers q = 1 + 4 r and 15 - q hold X[k] and X[M - k]; with (c, s) = (cos, sin)(2 pi k / N)
      // and P = s dr + c di, Q = c dr - s di:   a[k] = (sr - P, -(si + Q)),   a[M - k] = (sr + P, si - Q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 1 + 4 * r, qp = 15 - q;
        const float2 xk = xa[q], xm = xa[qp];
        const float2 w = s_twf[t + L + 4 * L * r];
        const float c = w.x, sn = -w.y;
        const float sr = xk.x + xm.x, si = xk.y - xm.y;
        const float dr = xk.x - xm.x, di = xk.y + xm.y;
        const float pp = fmaf(sn, dr, c * di), qq = fmaf(c, dr, -sn * di);
        a[q] = make_float2(sr - pp, -(si + qq));
        a[qp] = make_float2(sr + pp, si - qq);
      }
      // the outer groups pair across each other (q <-> 15 - q) -- except in lane 0, where group 0 pairs with itself
      // (bin 0 with the Nyquist bin) and so does group 128: one evaluation per register, partner by select
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = (i & 1) ? 4 * (i >> 1) + 3 : 4 * (i >> 1);
        float2 xk = xa[q];
        const float2 own = (q & 3) == 0 ? (q == 0 ? xN : xa[q == 0 ? 0 : 16 - q]) : xa[18 - q];
        float2 xm = t == 0 ? own : xa[15 - q];
        if (q == 0) {
          if (t == 0) { xk.y = 0.f; xm.y = 0.f; }   // c2r ignores the imaginary part of DC and Nyquist
        }
        const float2 w = s_twf[kq(q)];
        const float c = w.x, sn = -w.y;
        const float sr = xk.x + xm.x, si = xk.y - xm.y;
        const float dr = xk.x - xm.x, di = xk.y + xm.y;
        a[q] = make_float2(sr - sn * dr - c * di, -(si + c * dr - sn * di));
      }
    }
#pragma unroll
    for (int q = 0; q < (PAIRED ? 0 : 16); ++q) {
      const int k = t + L * q;
      float2 xk = xa[q];
      const float2 sv = make_float2(__shfl(xa[15 - q].x, src_lane, 64), __shfl(xa[15 - q].y, src_lane, 64));
      const float2 own = q == 0 ? xN : xa[q == 0 ? 0 : 16 - q];
      float2 xm = t == 0 ? own : sv;
      if constexpr (!ZERO_PAGE) { if (!live) { xk = make_float2(0.f, 0.f); xm = xk; } }
      if (k == 0) {  // c2r ignores the imaginary part of DC and Nyquist
        xk.y = 0.f; xm.y = 0.f;
        if (ADJ) { xk.x *= 2.f; xm.x *= 2.f; }
      }
      const float2 w = s_twf[k];
      const float c = w.x, s = -w.y;
      const float sr = xk.x + xm.x, si = xk.y - xm.y;
      const float dr = xk.x - xm.x, di = xk.y + xm.y;
      a[q] = make_float2(sr - s * dr - c * di, -(si + c * dr - s * di));
    }
    // reciprocal envelope of the segment this step finishes: registers (interior) or the L2-resident table
    float2 env[SH];
    if constexpr (ADJ) {
#pragma unroll
      for (int q = 0; q < SH; ++q) env[q] = make_float2(1.f, 1.f);
    } else {
      const int fe = f < 0 ? 0 : (f >= A.n_seg ? A.n_seg - 1 : f);
      const float2* __restrict__ e2 = reinterpret_cast<const float2*>(A.inv_env) + (int64_t)fe * (HOP / 2);
#pragma unroll
      for (int q = 0; q < SH; ++q) env[q] = e2[t + L * q];
    }
    live_nxt = issue_loads(f + 1);
    // The last pass of the transform runs in place on registers (pass_compute_regs): its outputs
    // are the points t + L q of this thread, the layout the register window needs -- no slab
    // round trip behind it.
    if constexpr (PAIRED) {
      // pass 1: the four radix-4 butterflies of this lane's groups, each written to points 4 j .. 4 j + 3
      const int jb[4] = {t, t + L, pj2, pj3};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float2 v[4] = {a[b], a[b + 4], a[b + 8], a[b + 12]};
        Dft<4>::run(v);
#pragma unroll
        for (int r = 0; r < 4; ++r) fbuf[phys<L>(4 * jb[b] + r)] = v[r];
      }
      wave_sync();
      load_points<L>(a, fbuf, t);
      wave_sync();
      float2 tw2[16], tw3[16];
      {
        const float2* rowp = reinterpret_cast<const float2*>(s_tw2 + (t & 3) * 36);
#pragma unroll
        for (int r = 1; r < 16; ++r) tw2[r] = rowp[r];
      }
      pass_compute_store<16, 4, L>(a, fbuf, t, tw2);
      wave_sync();
      load_points<L>(a, fbuf, t);
      wave_sync();
      if constexpr (R3P == 16) {
        const float2* rowp = reinterpret_cast<const float2*>(s_tw3 + t * 36);
#pragma unroll
        for (int r = 0; r < 16; ++r) tw3[r] = rowp[r];
        pass_compute_regs<R3P, 64, L>(a, tw3);
      } else {
        // M = 512: two radix-8 butterflies; of each butterfly's seven twiddles w, w^2, w^4 come from the table and the rest
        // are one or two products (all 14 entries read up front do not fit the register budget of the hop = n_fft / 4 kernel)
        constexpr int NBP = 16 / R3P;
        const float2* rowp = reinterpret_cast<const float2*>(s_tw3 + t * 36);
#pragma unroll
        for (int b = 0; b < NBP; ++b) {
          float2 v[R3P];
#pragma unroll
          for (int r = 0; r < R3P; ++r) v[r] = a[b + r * NBP];
          const float2 w1 = rowp[b * R3P + 1], w2 = rowp[b * R3P + 2], w4 = rowp[b * R3P + 4];
          const float2 w6 = cmul(w4, w2);
          v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], cmul(w2, w1)); v[4] = cmul(v[4], w4);
          v[5] = cmul(v[5], cmul(w4, w1)); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], cmul(w6, w1));
          Dft<R3P>::run(v);
#pragma unroll
          for (int r = 0; r < R3P; ++r) a[b + r * NBP] = v[r];
        }
      }
    } else if constexpr (P::R2 == 1) {
      pass_compute_regs<16, 1, L>(a, nullptr);
    } else {
      pass_compute_store<16, 1, L>(a, fbuf, t, nullptr);
      wave_sync();
      load_points<L>(a, fbuf, t);
      wave_sync();
      float2 tw2[NB2 * P::R2];
#pragma unroll
      for (int b = 0; b < NB2; ++b) {
        const float2* rowp = reinterpret_cast<const float2*>(s_tw2 + ((t + b * L) & 15) * 36);
#pragma unroll
        for (int r = 1; r < P::R2; ++r) tw2[b * P::R2 + r] = rowp[r];
      }
      if constexpr (P::R3 == 1) {
        pass_compute_regs<P::R2, 16, L>(a, tw2);
      } else {
        pass_compute_store<P::R2, 16, L>(a, fbuf, t, tw2);
        wave_sync();
        load_points<L>(a, fbuf, t);
        wave_sync();
        float2 tw3[NB3 * P::R3];
#pragma unroll
        for (int b = 0; b < NB3; ++b) {
          tw3[b * P::R3 + 1] = tw3b[b];
#pragma unroll
          for (int r = 2; r < P::R3; ++r) tw3[b * P::R3 + r] = cmul(tw3[b * P::R3 + r - 1], tw3b[b]);
        }
        pass_compute_regs<P::R3, 16 * P::R2, L>(a, tw3);
      }
    }
    // windowed frame into the register window: acc[q] covers samples 2(t + L q), +1 of frame f
    // (the window table carries the 1/N)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float2 w = s_win2[t + L * q];
      acc[q].x = fmaf(a[q].x, w.x, acc[q].x);
      acc[q].y = fmaf(-a[q].y, w.y, acc[q].y);
    }
    // segment f is complete: scale by 1/envelope, store, shift the window by one hop.
    // Interior segments (all lanes of the wave inside [0, length)) take straight-line stores; the
    // row edges go through the guarded path.
    {
      const int p_seg = f * HOP - (ADJ ? 0 : N / 2);           // first output sample of the segment
      const bool emit = unit_ok && f >= h0 && f < h1;
      if constexpr (DUMP) {
        // ONE straight-line sequence of SH pair stores on every path (an if / else between a fast and a guarded
        // form is turned into two flag-guarded regions by the structurizer, and the waitcnt pass then sees a
        // path with no store at all): pairs that are not entirely inside the row go to this lane's dump slot.
        bool cut = false;
#pragma unroll
        for (int q = 0; q < SH; ++q) {
          const int p = p_seg + 2 * (t + L * q);
          const bool full = emit && p >= 0 && p + 1 < len;
          cut |= emit && !full && p + 1 >= 0 && p < len;
          float* dst = full ? orow + p : A.dump + 128 * q + 2 * lane;   // one slot per q: stores to one address would be merged
          at::stg2_a4<AT_NT_ISTFT_ST != 0>(dst, acc[q].x * env[q].x, acc[q].y * env[q].y);
        }
        if (__any(cut)) {   // a pair cut by the end of an odd-length row (or by its start): element stores
#pragma unroll
          for (int q = 0; q < SH; ++q) {
            const int p = p_seg + 2 * (t + L * q);
            if (!emit || (p >= 0 && p + 1 < len)) continue;
            if (p >= 0 && p < len) orow[p] = acc[q].x * env[q].x;
            if (p + 1 >= 0 && p + 1 < len) orow[p + 1] = acc[q].y * env[q].y;
          }
        }
      } else {
        const bool inside = emit && p_seg >= 0 && p_seg + HOP <= len;
        if (__all(inside)) {
#pragma unroll
          for (int q = 0; q < SH; ++q) {
            at::stg2_a4<AT_NT_ISTFT_ST != 0>(orow + p_seg + 2 * (t + L * q), acc[q].x * env[q].x, acc[q].y * env[q].y);
          }
        } else if (emit) {
#pragma unroll
          for (int q = 0; q < SH; ++q) {
            const int p = p_seg + 2 * (t + L * q);
            if (p >= 0 && p < len) orow[p] = acc[q].x * env[q].x;
            if (p + 1 >= 0 && p + 1 < len) orow[p + 1] = acc[q].y * env[q].y;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 16 - SH; ++q) acc[q] = acc[q + SH];
#pragma unroll
    for (int q = 16 - SH; q < 16; ++q) acc[q] = make_float2(0.f, 0.f);
    // (renaming the window instead of shifting it -- the step unrolled over the N / hop phases -- was tried: the four
    //  copies of the step push the allocation past 256 registers, 93 spilled at M = 1024, 4.93 ms instead of 1.95)
  };

  // every slot runs the same trip count; frames outside [lead, lead + n_x) are all-zero
  const int total = A.run + R - 1;
  const int f_first = h0 - (R - 1);            // virtual frame of step 0
  step(f_first);   // peeled (see above)
  for (int i = 1; i < total; ++i) step(f_first + i);
}

template <int M, int SH, bool ADJ, bool MELB = false, bool EDIT = false>
int launch_fused(const IstftFusedArgs& A, hipStream_t stream) {
  constexpr int FW = Plan<M>::FW;
  const int64_t waves = (A.total_units + FW - 1) / FW;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return AT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((istft_fused_kernel<M, SH, ADJ, MELB, EDIT>), dim3((unsigned)blocks), dim3(256), 0, stream, A);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

template <int M, bool ADJ>
int launch_fused_sh(int sh, const IstftFusedArgs& A, hipStream_t stream) {
  switch (sh) {
    case 1: return launch_fused<M, 1, ADJ>(A, stream);
    case 2: return launch_fused<M, 2, ADJ>(A, stream);
    case 4: return launch_fused<M, 4, ADJ>(A, stream);
    case 8: return launch_fused<M, 8, ADJ>(A, stream);
  }
  return AT_ERR_UNSUPPORTED;
}

// fused STFT-domain edit: hop = n_fft / 4 (the default hop of every SpectralTransform), n_fft >= 64
int launch_fused_edit(int M, const IstftFusedArgs& F, hipStream_t st) {
  switch (M) {
    case 32: return launch_fused<32, 4, false, false, true>(F, st);
    case 64: return launch_fused<64, 4, false, false, true>(F, st);
    case 128: return launch_fused<128, 4, false, false, true>(F, st);
    case 256: return launch_fused<256, 4, false, false, true>(F, st);
    case 512: return launch_fused<512, 4, false, false, true>(F, st);
    case 1024: return launch_fused<1024, 4, false, false, true>(F, st);
  }
  return AT_ERR_UNSUPPORTED;
}

// mel backward: hop = n_fft/4 only (the hop every mel loss of the reference uses), n_fft >= 64
int launch_fused_melb(int M, const IstftFusedArgs& F, hipStream_t st) {
  switch (M) {
    case 32: return launch_fused<32, 4, true, true>(F, st);
    case 64: return launch_fused<64, 4, true, true>(F, st);
    case 128: return launch_fused<128, 4, true, true>(F, st);
    case 256: return launch_fused<256, 4, true, true>(F, st);
    case 512: return launch_fused<512, 4, true, true>(F, st);
    case 1024: return launch_fused<1024, 4, true, true>(F, st);
  }
  return AT_ERR_UNSUPPORTED;
}

template <bool ADJ>
int launch_fused_m(int M, int sh, const IstftFusedArgs& F, hipStream_t st) {
  switch (M) {
    case 16: return launch_fused_sh<16, ADJ>(sh, F, st);
    case 32: return launch_fused_sh<32, ADJ>(sh, F, st);
    case 64: return launch_fused_sh<64, ADJ>(sh, F, st);
    case 128: return launch_fused_sh<128, ADJ>(sh, F, st);
    case 256: return launch_fused_sh<256, ADJ>(sh, F, st);
    case 512: return launch_fused_sh<512, ADJ>(sh, F, st);
    case 1024: return launch_fused_sh<1024, ADJ>(sh, F, st);
  }
  return AT_ERR_UNSUPPORTED;
}

// run partition shared by the inverse and the adjoint
void plan_runs(IstftFusedArgs& F, int64_t rows, int n_fft, int hop) {
  const int R = n_fft / hop;
  F.n_seg = F.n_frames - 1 + R;
  // A unit is one (row, run of segments) and costs run + R - 1 frames (R - 1 warm-up frames are
  // transformed again at every run boundary).  Units run in ROUNDS of one per resident frame slot
  // (2 waves per SIMD, 1024 / M frames per wave), so the launch lasts  rounds x (run + R - 1)  frame
  // times: take the runs-per-row that minimises it.  (B = 512 x 2 ch x 10 s: 2 runs of 433
  // segments = exactly one round; the former fixed target of ~16k units, 16 runs of 55, spent 5 %
  // of the launch on warm-up frames and ran 3.7 % slower, profiles/r02_notes.md.)
  static const int want_units = at::env_int_once("AT_ISTFT_UNITS", 0);   // measurement knob: fixed unit target
  int run;
  if (want_units > 0) {
    const int64_t want = (want_units + rows - 1) / rows;
    run = (int)((F.n_seg + want - 1) / want);
    if (run < 8 * R) run = 8 * R;
  } else {
    const int fw = n_fft >= 2048 ? 1 : 2048 / n_fft;
    const int64_t slots = (int64_t)at::device_cu_count() * 4 * AT_ISTFT_WPS * fw;
    const int max_rp = F.n_seg / (8 * R) > 1 ? F.n_seg / (8 * R) : 1;
    double best_cost = 1e300;
    run = F.n_seg;
    for (int rp = 1; rp <= max_rp && rp <= 4096; ++rp) {
      const int r = (F.n_seg + rp - 1) / rp;
      const int64_t units = rows * ((F.n_seg + r - 1) / r);
      const int64_t rounds = (units + slots - 1) / slots;
      const double cost = (double)rounds * (double)(r + R - 1);
      if (cost < best_cost * 0.999) { best_cost = cost; run = r; }
    }
  }
  if (run > F.n_seg) run = F.n_seg;
  if (run < 1) run = 1;
  F.run = run;
  F.runs_per_row = (F.n_seg + run - 1) / run;
  F.total_units = rows * F.runs_per_row;
}

// hop = n_fft / {2,4,8,16}  <=>  hop = 2 L SH with SH in {8,4,2,1}
int fused_shift(int n_fft, int hop) {
  for (int r = 2; r <= 16; r *= 2)
    if (hop * r == n_fft) return 16 / r;
  return 0;
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

// bytes of workspace: the reciprocal envelope ((n_frames-1)*hop + n_fft floats) for the fused path,
// plus the (rows, n_frames, n_fft) float frame buffer when hop is not n_fft / {2,4,8,16}
int64_t at_istft_workspace_bytes(int64_t rows, int64_t n_frames, int n_fft, int hop) {
  if (rows < 0 || n_frames <= 0 || n_fft <= 0 || hop <= 0) return AT_ERR_INVALID;
  const bool fused_size = n_fft >= 32 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0;
  if (fused_size && fused_shift(n_fft, hop))     // envelope table + dump slots + zero page
    return ((n_frames - 1) * hop + n_fft + DUMP_FLOATS + ZERO_PAGE_FLOATS) * 4;
  static const int tiled_off = at::env_int_once("AT_ISTFT_TILED_OFF", 0);       // A/B: frame buffer + gather for 4096 / 8192
  if (!tiled_off && at::istft_tiled_supported(n_fft, hop)) return at::istft_tiled_workspace_floats(n_frames, n_fft, hop) * 4;
  if (at::istft_generic_ola_supported(n_fft, hop)) return at::istft_generic_ola_workspace_floats(n_frames, n_fft, hop) * 4;
  return rows * n_frames * (int64_t)n_fft * 4;
}

// X (rows, n_x, n_fft/2+1) complex64 interleaved, bin-contiguous (as at_stft_mel_f32 writes it).
// The transform is taken over n_frames = lead + n_x + trail VIRTUAL frames of which the first
// `lead` and the last n_frames - lead - n_x are all-zero (match_stride re-inserts the two edge
// frames the forward transform dropped, audio_signal.py:1278-1281, without copying X).
// out (rows, length): sample p comes from centre-padded position p + n_fft/2.
static int istft_run(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft,
                     int hop, int lead, int64_t n_frames, int64_t length, float* out, void* workspace,
                     int64_t workspace_bytes, void* stream, const IstftFusedArgs* edit) {
  if (rows == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!X || !window || !twiddles || !out || rows < 0 || n_x <= 0 || hop <= 0 || length < 0 || lead < 0 ||
      n_frames < lead + n_x)
    return AT_ERR_INVALID;
  const bool fused_size = n_fft >= 32 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0;
  if (n_frames >= (1LL << 31) / (hop > n_fft ? hop : n_fft) || length >= (1LL << 31) - n_fft) return AT_ERR_UNSUPPORTED;
  if (rows == 0 || length == 0) return AT_OK;
  if (!workspace || workspace_bytes < at_istft_workspace_bytes(rows, n_frames, n_fft, hop)) return AT_ERR_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!fused_size) {
    if (edit) return AT_ERR_UNSUPPORTED;
    static const int tiled_off = at::env_int_once("AT_ISTFT_TILED_OFF", 0);
    if (!tiled_off && at::istft_tiled_supported(n_fft, hop))
      return at::istft_tiled(X, rows, n_x, window, twiddles, n_fft, hop, lead, n_frames, length, out,
                             reinterpret_cast<float*>(workspace), st);
    // generic sizes (4096 ..., non powers of two): mixed-radix frames + the gather kernel
    if (lead != 0 || n_frames != n_x) return AT_ERR_UNSUPPORTED;
    if (at::istft_generic_ola_supported(n_fft, hop))      // one pass: overlap-add in LDS, no frame buffer
      return at::istft_generic_ola(X, rows, n_frames, window, twiddles, n_fft, hop, length, out,
                                   reinterpret_cast<float*>(workspace), st);
    int rc = at::istft_frames_generic(X, rows, n_frames, window, twiddles, n_fft, reinterpret_cast<float*>(workspace), st);
    if (rc != AT_OK) return rc;
    const int64_t total = rows * length;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float*>(workspace),
                       window, out, rows, (int)n_frames, n_fft, hop, length);
    AT_LAUNCH_CHECK();
    return AT_OK;
  }
  const int M = n_fft / 2;
  const int FW = 64 / (M / 16);
  const int sh = fused_shift(n_fft, hop);
  {
    // n_fft <= 128 take the generic one-pass kernel (consecutive frames per tile, overlap-add in LDS) instead of the fused one,
    // whose 16 / 32 frame slots per wave walk different row segments: its loads are 16- and 32-byte pieces there.  Round 6, s09,
    // B = 512 x 2 ch x 10 s @ 8 kHz, hop = n_fft / 4: n_fft 64: 4.89 -> 1.22 ms, 128: 1.32 -> 1.10 ms; 256 / 512 stay fused (0.42 /
    // 0.35 against 1.08 ms).  (AT_ISTFT_SMALL_OLA = largest n_fft routed this way: development A/B.)
    static const int small_ola = at::env_int_once("AT_ISTFT_SMALL_OLA", 128);
    if (small_ola && n_fft <= small_ola && !edit && lead == 0 && n_frames == n_x && at::istft_generic_ola_supported(n_fft, hop))
      return at::istft_generic_ola(X, rows, n_frames, window, twiddles, n_fft, hop, length, out,
                                   reinterpret_cast<float*>(workspace), st);
  }
  if (sh) {
    IstftFusedArgs F;
    F.X = reinterpret_cast<const float2*>(X); F.window = window; F.tw = reinterpret_cast<const float2*>(twiddles);
    F.inv_env = reinterpret_cast<const float*>(workspace); F.out = out; F.rows = rows; F.length = length;
    F.dump = reinterpret_cast<float*>(workspace) + ((n_frames - 1) * hop + n_fft);
    F.zeros = reinterpret_cast<const float2*>(F.dump + DUMP_FLOATS);
    F.n_x = (int)n_x; F.lead = lead; F.n_frames = (int)n_frames;
    F.gmel = nullptr; F.bin_bands = nullptr; F.bin_w = nullptr; F.n_mels = 0;
    F.edit_kind = 0;
    if (edit) {
      if (sh != 4 || M < 32) return AT_ERR_UNSUPPORTED;
      F.edit_kind = edit->edit_kind; F.edit_C = edit->edit_C; F.edit_lo = edit->edit_lo; F.edit_hi = edit->edit_hi;
      F.edit_shift = edit->edit_shift; F.edit_cut = edit->edit_cut; F.edit_maxpow = edit->edit_maxpow;
      F.edit_fill = edit->edit_fill; F.edit_top_db = edit->edit_top_db; F.edit_val = edit->edit_val;
      F.edit_use_top = edit->edit_use_top;
    }
    plan_runs(F, rows, n_fft, hop);
    const int64_t env_n = (n_frames - 1) * hop + n_fft;
    int64_t eb = (env_n + 255) / 256;
    if (eb > 4096) eb = 4096;
    hipLaunchKernelGGL(istft_env_kernel, dim3((unsigned)eb), dim3(256), 0, st, window, reinterpret_cast<float*>(workspace),
                       (int)n_frames, n_fft, hop, env_n);
    AT_LAUNCH_CHECK();
    return edit ? launch_fused_edit(M, F, st) : launch_fused_m<false>(M, sh, F, st);
  }
  if (edit) return AT_ERR_UNSUPPORTED;
  // generic hop: frame buffer + gather.  Virtual zero frames are not supported here.
  if (lead != 0 || n_frames != n_x) return AT_ERR_UNSUPPORTED;
  if (at::istft_generic_ola_supported(n_fft, hop))        // even hops: transform tiles + overlap-add in LDS (workspace = envelope)
    return at::istft_generic_ola(X, rows, n_frames, window, twiddles, n_fft, hop, length, out,
                                 reinterpret_cast<float*>(workspace), st);
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

int at_istft_f32(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft,
                 int hop, int lead, int64_t n_frames, int64_t length, float* out, void* workspace,
                 int64_t workspace_bytes, void* stream) {
  return istft_run(X, rows, n_x, window, twiddles, n_fft, hop, lead, n_frames, length, out, workspace, workspace_bytes,
                   stream, nullptr);
}

// at_istft_f32 with an STFT-domain edit applied to X as it is read (X itself is not modified): the inverse transform
// of  edit(X)  for the edits of dsp.py:217-352 with per-ITEM parameters (rows = B * C, item = row / C):
//   kind 1  mask_frequencies: bins  lo[b] <= k < hi[b]  of every frame := (fill_re, fill_im)
//   kind 2  mask_timesteps:   frames lo[b] <= n < hi[b] := (fill_re, fill_im)
//   kind 3  shift_phase:      X * e^{i shift[b]}
//   kind 4  mask_low_magnitudes: 10 log10(max(|X|^2, 1e-10)) (floored at max - top_db when use_top_db; maxpow = the
//           batch maximum of |X|^2 from at_spec_maxpow_f32) < cut_db[b]  ->  val e^{i angle X}
// The integer ranges are the reference's float comparisons against the (monotone) frequency / time grid, evaluated
// by the caller.  hop must be n_fft / 4 and 64 <= n_fft <= 2048 (AT_ERR_UNSUPPORTED otherwise: apply the edit with
// at_spec_* and call at_istft_f32).
int at_istft_edit_f32(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft,
                      int hop, int lead, int64_t n_frames, int64_t length, float* out, void* workspace,
                      int64_t workspace_bytes, int kind, int64_t C, const int* lo, const int* hi, const float* shift,
                      const double* cut_db, const float* maxpow, float fill_re, float fill_im, float top_db,
                      int use_top_db, float val, void* stream) {
  if (kind < 1 || kind > 4 || C <= 0 || rows % C != 0) return AT_ERR_INVALID;
  if ((kind <= 2 && (!lo || !hi)) || (kind == 3 && !shift) || (kind == 4 && (!cut_db || !maxpow))) return AT_ERR_INVALID;
  IstftFusedArgs E;
  E.edit_kind = kind; E.edit_C = (int)C; E.edit_lo = lo; E.edit_hi = hi; E.edit_shift = shift; E.edit_cut = cut_db;
  E.edit_maxpow = reinterpret_cast<const unsigned*>(maxpow); E.edit_fill = make_float2(fill_re, fill_im);
  E.edit_top_db = top_db; E.edit_val = val; E.edit_use_top = use_top_db;
  return istft_run(X, rows, n_x, window, twiddles, n_fft, hop, lead, n_frames, length, out, workspace, workspace_bytes,
                   stream, &E);
}

// Adjoint of the forward STFT (the backward pass of stft() for a real signal):
//   G (rows, n_frames, n_fft/2+1) complex64 = dL/dX as autograd hands it, bin-contiguous
//   out (rows, out_len), out_len >= Lp = (n_frames-1)*hop + n_fft:
//       out[n] = sum_f window[n - f hop] * sum_k Re(G[f,k] e^{+2 pi i k (n - f hop)/n_fft})   for n < Lp
// (positions >= Lp are not written), i.e. the gradient w.r.t. the CENTRE-PADDED signal; the caller
// folds the reflected margins back (audio_signal.py:1195 torch.stft(center=True)).
// hop must be n_fft / {2,4,8,16}.
int at_stft_adjoint_f32(const float* G, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                        int n_fft, int hop, float* out, int64_t out_len, void* stream) {
  if (rows == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!G || !window || !twiddles || !out || rows < 0 || n_frames <= 0 || hop <= 0 ||
      out_len < (n_frames - 1) * hop + n_fft || out_len >= (1LL << 31))
    return AT_ERR_INVALID;
  if (!(n_fft >= 32 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0)) return AT_ERR_UNSUPPORTED;
  const int sh = fused_shift(n_fft, hop);
  if (!sh) return AT_ERR_UNSUPPORTED;
  if (n_frames >= (1LL << 31) / n_fft) return AT_ERR_UNSUPPORTED;
  if (rows == 0) return AT_OK;
  IstftFusedArgs F;
  F.X = reinterpret_cast<const float2*>(G); F.window = window; F.tw = reinterpret_cast<const float2*>(twiddles);
  F.inv_env = nullptr; F.dump = nullptr; F.zeros = nullptr; F.out = out; F.rows = rows; F.length = out_len;
  F.n_x = (int)n_frames; F.lead = 0; F.n_frames = (int)n_frames;
  F.gmel = nullptr; F.bin_bands = nullptr; F.bin_w = nullptr; F.n_mels = 0;
  plan_runs(F, rows, n_fft, hop);
  return launch_fused_m<true>(n_fft / 2, sh, F, reinterpret_cast<hipStream_t>(stream));
}

// Backward of the fused STFT + mel path w.r.t. the signal, given ONLY dL/dmel:
//   X     (rows, n_frames, n_fft/2+1) complex64: the spectrum the forward pass stored
//   gmel  (rows, n_frames, n_mels) f32: dL/dmel, band-contiguous
//   bin_bands (n_fft/2+1) i32 = lo | hi << 16, bin_w (n_fft/2+1, 2) f32: per bin the (at most two)
//         bands of the triangular bank that cover it and their weights (0 where absent)
//   out   as at_stft_adjoint_f32.  hop must be n_fft/4, 64 <= n_fft <= 2048, n_mels <= n_fft/8.
int at_stft_mel_adjoint_f32(const float* X, const float* gmel, const int* bin_bands, const float* bin_w, int n_mels,
                            int64_t rows, int64_t n_frames, const float* window, const float* twiddles, int n_fft,
                            int hop, float* out, int64_t out_len, void* stream) {
  if (rows == 0) return AT_OK;
  if (!X || !gmel || !bin_bands || !bin_w || !window || !twiddles || !out || rows < 0 || n_frames <= 0 || n_mels <= 0 ||
      out_len < (n_frames - 1) * (int64_t)hop + n_fft || out_len >= (1LL << 31))
    return AT_ERR_INVALID;
  if (!(n_fft >= 64 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0) || hop * 4 != n_fft) return AT_ERR_UNSUPPORTED;
  if (n_mels > 8 * (n_fft / 32) || n_mels >= 0xffff || n_frames >= (1LL << 31) / n_fft) return AT_ERR_UNSUPPORTED;
  IstftFusedArgs F;
  F.X = reinterpret_cast<const float2*>(X); F.window = window; F.tw = reinterpret_cast<const float2*>(twiddles);
  F.inv_env = nullptr; F.dump = nullptr; F.zeros = nullptr; F.out = out; F.rows = rows; F.length = out_len;
  F.n_x = (int)n_frames; F.lead = 0; F.n_frames = (int)n_frames;
  F.gmel = gmel; F.bin_bands = bin_bands; F.bin_w = reinterpret_cast<const float2*>(bin_w); F.n_mels = n_mels;
  plan_runs(F, rows, n_fft, hop);
  return launch_fused_melb(n_fft / 2, F, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
