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
//    between passes goes through a per-wave LDS slab of 8704 B (bank-swizzled,
//    see phys<L>() in fft_wave.h).  Waves never synchronise with each other
//    (no s_barrier after the table setup);
//  * a wave walks runs of consecutive frames of one row inside its XCD's span, so
//    the 4x overlap of the input between frames is served by registers (hop =
//    n_fft/4: shift by 4) and the XCD's L2; HBM sees each sample once;
//  * window, split twiddles, pass-2 twiddles and the mel tables live in LDS
//    (per block), pass-3 twiddles are derived from one base twiddle per butterfly;
//  * global loads are float2 per lane, 512 B contiguous per wave instruction;
//    global stores are float2 per lane, 512 B contiguous;
//  * mel: the Slaney filterbank is banded (each bin feeds <= 2 bands), so it is
//    applied as "units" of 16 bins x 1 band from an LDS copy of |X| instead of a
//    dense (F x n_mels) GEMM: 2 flop/bin instead of 2*n_mels flop/bin.
#include "at_common.h"
#include "fft_wave.h"
#include "generic_fft.h"
#include <stdlib.h>
#include <type_traits>

// build-time tuning knobs (see DESIGN.md "STFT kernel tuning")
#ifndef AT_STFT_NW
#define AT_STFT_NW 4            // waves per workgroup
#endif
#ifndef AT_STFT_WPS
#define AT_STFT_WPS 2           // resident waves per SIMD the register budget allows
#endif
#ifndef AT_STFT_WPS_SMALL
#define AT_STFT_WPS_SMALL 2     // the same for n_fft <= 512 (their 152-164 registers would allow 3)
#endif
#ifndef AT_STFT_RUNSTORE
#define AT_STFT_RUNSTORE 1      // 0 = A/B build: the several-frames-per-wave kernels store FW segments per instruction instead of 512-byte runs from the slab
#endif
#ifndef AT_STFT_RUNSTORE_NT
#define AT_STFT_RUNSTORE_NT 1   // those runs as non-temporal stores (0 = A/B build).  With 512-byte runs `nt` pays as it does in the v2 kernel: n_fft 512
                                // 0.81-0.87 -> 0.70 ms, 1024 @ 44.1 kHz 2.06 -> 1.93, 256: 0.42-0.47 -> 0.39 (s33); on the old 128-byte segments it had
                                // cost 25 % (they lost the L2's write combining, r04_notes.md 12)
#endif
#ifndef AT_STFT_SEG_NT
#define AT_STFT_SEG_NT 1        // the 256-byte segments of n_fft 1024 + mel (the one several-frames-per-wave case left on the old store path) non-temporal:
                                // 2.59 -> 2.50 ms at 44.1 kHz, 1.27 -> 1.24 at 22 kHz (s34); 0 = A/B build
#endif
#ifndef AT_STFT_MELREG
#define AT_STFT_MELREG 1        // 0 = A/B build: unit descriptors (and two rounds' weights) of the several-frames-per-wave kernels re-read from LDS per frame
#endif
#ifndef AT_STFT_MELUNROLL
#define AT_STFT_MELUNROLL 1     // A/B build: unroll factor of the per-frame mel loop of those kernels (2: within 1 %)
#endif
#ifndef AT_STFT_PIPE
#define AT_STFT_PIPE 1           // A/B build: 0 = the groups of the n_fft <= 1024 kernels load behind their predecessor's stores
#endif
#ifndef AT_STFT_DEBUGMODES
#define AT_STFT_DEBUGMODES 0     // 1: honour AT_STFT_DEBUG=1 (no stores) / 2 (store only) at run time
#endif
#ifndef AT_STFT_STAGGER
#define AT_STFT_STAGGER 0       // start-up stagger per wave slot, in units of 64 cycles (0 = off)
#endif
#ifndef AT_STFT_RUN
#define AT_STFT_RUN 16          // consecutive frame groups a wave handles before jumping ahead
#endif
#ifndef AT_STFT_V2_PIPE
#define AT_STFT_V2_PIPE 3       // v2 kernel: the mel unit rounds of frame f run inside frame f + 1, beside the register-only part of its first pass.
                                // 0 = the round-5 kernel (A/B build); 1 = placement left to the compiler, 2 = five slots held by scheduling barriers,
                                // two operand buffers (both spill: 256 registers + 23-38 scratch dwords, whose reloads count in vmcnt);
                                // 3 = the slots with ONE operand buffer: 252 registers, no scratch.  Same box, interleaved, placed outputs
                                // (profiles/sessions_r06/s01, s02): 1.988-2.032 -> 1.934-1.972 ms
#endif
#if !AT_DEV_KNOBS
#undef AT_STFT_ABL              // measurement builds exist in the development library only
#endif
#ifndef AT_STFT_ABL
#define AT_STFT_ABL 0           // measurement builds of the v2 kernel (results WRONG): 1 no v_sqrt, 2 no unit rounds, 4 no magnitude writes, 8 no thread-0
                                // selects, 16 no partner permute.  What they measured (s02, ms against 1.934-1.961): 1.94 / 1.77-1.82 / 1.89-1.90 / 1.93-1.94 /
                                // 1.93: the kernel pays for LDS traffic (the 32 ds_read_b128 of the unit rounds, the 16 magnitude writes), not for VALU
                                // instructions (33 selects: nothing; 16 v_sqrt: nothing)
#endif
#ifndef AT_STFT_V2_HX
#define AT_STFT_V2_HX 0         // A/B build (1): the ascending butterflies of pass 3 take their inputs by v_permlane16/32_swap instead of through the
                                // slab (8 ds_write_b64 + 8 ds_read_b64 fewer per frame, 16 swaps more).  Bit-compatible results (60 parity tests);
                                // same box, interleaved, three rounds: 1.958 / 1.963 / 1.965 vs 1.969 / 1.960 / 1.916 ms -- a draw (session s12)
#endif
#ifndef AT_STFT_RUN_V2
#define AT_STFT_RUN_V2 72       // the same for the 2048/512 kernel (runs of 36...431 measure alike, 16 is 3 % slower)
#endif

namespace {

struct StftArgs {
  const float* x;          // (rows, T)
  const float* window;     // (n_fft)
  const float2* tw;        // (n_fft): (cos, -sin)(2 pi k / n_fft)
  float2* out;             // (rows, n_out, M+1) or null
  float* mel;              // (rows, n_out, n_mels) or null
  const int* unit_info;    // (n_units_padded, 2): {row16 | mel << 16, shuffle mask}; see below
  const float* unit_w;     // (n_units_padded, 16)
  int64_t T;
  int64_t rows;
  int64_t n_out;           // frames written per row
  int64_t T2;              // T + 2*pad + right_pad
  int frame_lo;            // first frame computed (2 when match_stride drops edges)
  int hop;
  int pad;                 // outer left pad (match_stride)
  int pad_mode;
  int groups_per_row;      // ceil(n_out / FW)
  int64_t total_groups;    // rows * groups_per_row
  int n_units;             // padded to a multiple of 64
  int n_mels;
  int reuse_shift;         // hop / (2 L) when consecutive frames of a wave can reuse registers, else 0
  int run;                 // consecutive frame groups a wave handles before jumping ahead (<= AT_STFT_RUN)
  int debug;               // development: 1 = compute but never store, 2 = store only (no FFT)
  int flags;               // measurement knob AT_STFT_FLAGS (read once): cache policy of the v2 kernel's streaming traffic (POL)
  int run_max;             // upper bound of `run` for the v2 kernel (AT_STFT_RUNMAX, read once)
  int n_xcd;               // spans the v2 schedule cuts the frame range into (CUs / 32, i.e. 8 on a whole MI355X; AT_STFT_NX overrides)
  int stagger;             // v2: start-up delay per wave slot of a CU, in units of 64 cycles (AT_STFT_STAGGERV2)
};

using at::fetch_padded;
template <int POL>
__device__ __forceinline__ void st2(float2* p, float2 v);   // (defined with the v2 kernel below: bit 0 of POL = non-temporal)

// ---- mel "unit" tables (built by at_mel_units_host) --------------------------------------
// The Slaney filterbank is banded: band m is non-zero on a short run of bins.  Bins are cut
// into rows of 16 (row r = bins 16r..16r+15); a UNIT is one (row, band) pair with its 16
// weights.  Units are sorted by band, padded so that no band straddles a 64-lane round, and
// every lane of a round owns one unit:   acc = sum_i w[i] * |X|[16 row + i]   (4+4 ds_read_b128)
// The units of a band sit in adjacent lanes; they are summed with 3 shuffle-down steps whose
// participation bits come from the table, and the first lane of each band stores the result.
//   unit_info[2u]   = row16 | (mel << 16)       (mel = 0xffff for padding units)
//   unit_info[2u+1] = bit0..2: add lane+1 / +2 / +4 ;  bit3: this lane stores band `mel`
constexpr int MAG_ROW = 20;   // floats per padded LDS row of 16 magnitudes (80 B: conflict-free b128)
constexpr int MELW_ROW = 20;  // floats per padded LDS row of 16 unit weights

// LDS layout of one block (floats):
//   [NW wave slabs: NW * 2 * WAVE_LDS_SLOTS][window: 2M][split twiddles: M (= M/2 float2)]
//   [pass-2 twiddles: 16 rows (j mod 16) x 36 floats (16 float2 + pad, conflict-free b128 rows)]
//   [mel unit weights: n_units x MELW_ROW][mel unit info: n_units x int2 {row offset, band | flags}]
template <int M>
__host__ __device__ constexpr int lds_fixed_floats(int nw) { return nw * 2 * WAVE_LDS_SLOTS + 2 * M + M + 16 * 36; }

// 512-byte-run stores (AT_STFT_RUNSTORE): every size with several frames per wave; with the mel stage only where the extra
// magnitude region (NW x FW x (M/16 + 1) x MAG_ROW floats, 21-25 KB) still leaves two workgroups per CU: n_fft 128 ... 512
// with two rounds of units (80 mel bands: 73-75 KB per workgroup).  n_fft 1024 with mel measured no gain (s24).
template <int M, int NR>
__host__ __device__ constexpr bool runstore_for() { return AT_STFT_RUNSTORE && M <= 512 && (NR == 0 || (NR == 2 && M >= 64 && M <= 256)); }
template <int M, int NR>
__host__ __device__ constexpr int runstore_mag_floats(int nw) {
  return (runstore_for<M, NR>() && NR > 0) ? nw * (64 / (M / 16)) * (M / 16 + 1) * MAG_ROW : 0;
}

template <int M, int NW, bool VEC2, int NR /* mel rounds of 64 units; 0 = no mel */>
__global__ __launch_bounds__(NW * 64, (M <= 256 ? AT_STFT_WPS_SMALL : AT_STFT_WPS)) void stft_mel_kernel(const StftArgs A) {
  constexpr bool MEL = NR > 0;
  using P = Plan<M>;
  constexpr int L = P::L, FW = P::FW, N = 2 * M;
  constexpr int MAG_ROWS = M / 16 + 1;            // rows of 16 bins per frame (last row: Nyquist only)
  // FW == 1: one frame per wave, every scheduled frame exists -> stores are unconditional
  // straight-line code (the waitcnt pass can then count them behind the prefetched loads).
#if AT_STFT_DEBUGMODES
#define STORE_OK(v) (live && !(A.debug == 1 && (v) != 12345.678f))
#else
#define STORE_OK(v) true        // every frame slot of a wave holds a real frame (slots past the row's last frame repeat it)
#endif
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float* s_win = lds_f + NW * 2 * WAVE_LDS_SLOTS;
  float2* s_twp = reinterpret_cast<float2*>(s_win + N);
  float* s_tw2 = s_win + N + M;
  float* s_melw = s_tw2 + 16 * 36;
  int2* s_uinfo = reinterpret_cast<int2*>(s_melw + A.n_units * MELW_ROW);
  [[maybe_unused]] float* s_mag = reinterpret_cast<float*>(s_uinfo + A.n_units);   // RUNSTORE with mel: NW x FW x MAG_ROWS x MAG_ROW

  // ---- block-shared tables -> LDS (once per persistent block)
  // the window is stored HALVED: the factor 1/2 of the split step (X = (Z + conj Z')/2 ...) is exact
  // in binary floating point wherever it is applied, so it rides on the window multiply for free
  for (int i = threadIdx.x; i < N; i += NW * 64) s_win[i] = 0.5f * A.window[i];
  for (int i = threadIdx.x; i < M / 2; i += NW * 64) s_twp[i] = A.tw[i];
  if constexpr (P::R2 > 1) {
    // row = j mod 16, column r: exp(-2 pi i r (j mod 16) / (16 R2))
    for (int i = threadIdx.x; i < 16 * P::R2; i += NW * 64) {
      const int jj = i / P::R2, r = i % P::R2;
      reinterpret_cast<float2*>(s_tw2 + jj * 36)[r] = A.tw[r * jj * (N / (16 * P::R2))];
    }
  }
  if constexpr (MEL) {
    for (int i = threadIdx.x; i < A.n_units * 16; i += NW * 64)
      s_melw[(i >> 4) * MELW_ROW + (i & 15)] = A.unit_w[i];
    for (int i = threadIdx.x; i < A.n_units; i += NW * 64) {
      const int info = A.unit_info[2 * i], fl = A.unit_info[2 * i + 1];
      // x: float offset of the unit's row of magnitudes; y: flags | band << 8 (band 0xffff = none)
      s_uinfo[i] = make_int2((info & 0xffff) * MAG_ROW, (fl & 0xff) | (((info >> 16) & 0xffff) << 8));
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fs = lane / L;  // frame slot inside the wave
  const int t = lane % L;   // thread inside the frame

  float2* wbuf = reinterpret_cast<float2*>(lds_f) + wave * WAVE_LDS_SLOTS;
  float2* fbuf = wbuf + fs * P::SLOTS;

  // ---- per-thread constants
  constexpr int NB2 = 16 / P::R2;
  constexpr int NB3 = 16 / P::R3;
  float2 tw3b[P::R3 > 1 ? NB3 : 1];  // base twiddle w^1 per butterfly of pass 3; w^2, w^3.. derived
  if constexpr (P::R3 > 1) {
    constexpr int NS = 16 * P::R2;
#pragma unroll
    for (int b = 0; b < NB3; ++b) {
      const int j = t + b * L;
      tw3b[b] = A.tw[(j % NS) * (N / (NS * P::R3))];
    }
  }
  // Several frames per wave: the mel stage runs FW times per group with the same units in every lane -- their descriptors
  // (and, for two rounds, their sixteen weights each) stay in registers instead of being re-read from LDS per frame.
  constexpr bool MELDESC = MEL && FW > 1 && AT_STFT_MELREG;
  constexpr bool MELREG = MELDESC && NR <= 2 && M <= 256;    // (n_fft 1024: 24 bytes of scratch per lane with them)
  [[maybe_unused]] int mu_off[MELDESC ? NR : 1], mu_fl[MELDESC ? NR : 1];
  [[maybe_unused]] float4 mu_w[MELREG ? NR : 1][4];
  if constexpr (MELDESC) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int2 ui = s_uinfo[r * 64 + lane];
      mu_off[r] = ui.x;
      mu_fl[r] = ui.y;
      if constexpr (MELREG) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) mu_w[r][i4] = reinterpret_cast<const float4*>(s_melw + (r * 64 + lane) * MELW_ROW)[i4];
      }
    }
  }
  const int Ti = (int)A.T;
  const int n_out = (int)A.n_out;
  const int gpr = A.groups_per_row;
  // Persistent schedule.  Workgroup b runs on XCD b % 8 (observed placement; used for speed
  // only).  The (row, frame-group) space is cut into one CONTIGUOUS span per XCD and the waves
  // of an XCD interleave inside their span: the 4x overlap between neighbouring frames is then
  // served by that XCD's own L2 (HBM reads each sample once), and each XCD writes one compact
  // moving window of the output.
  const int n_x = gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % n_x;
  const int lblk = blockIdx.x / n_x;
  const int nblk_x = ((int)gridDim.x - xcd + n_x - 1) / n_x;       // blocks on this XCD
  const int64_t g_lo = A.total_groups * xcd / n_x;
  const int64_t g_hi = A.total_groups * (xcd + 1) / n_x;
  const int RUN = A.run;
  const int64_t Wtot = (int64_t)nblk_x * NW;                        // waves working on this span
  const int64_t w0 = g_lo + ((int64_t)lblk * NW + wave) * RUN;      // first group of this wave

  // De-phase the persistent waves: they all start together and execute identical work, so
  // without a stagger the whole chip alternates between "everyone transforms" and "everyone
  // stores" and the memory system idles half of the time.  Eight slots per CU (2 blocks x 4
  // waves) spread over roughly one frame time.
  if constexpr (AT_STFT_STAGGER > 0) {
    const int slot = (wave + NW * (lblk & 1)) & 7;
    for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(AT_STFT_STAGGER);
  }
  constexpr int SH = 4;    // register shift between consecutive frames (hop = n_fft / 4)
  const bool can_reuse = (FW == 1) && A.reuse_shift == SH;
  float2 raw[16];          // un-windowed samples of the frame being processed (register reuse)
  float2 nxt[SH];          // the SH new loads of the NEXT frame, issued ahead of this frame's stores
  // FW > 1 (n_fft <= 1024: 2 ... 32 frames per wave, no overlap between consecutive groups of a wave): ALL sixteen loads
  // of the next group, issued BEFORE this group's transform (a whole group of lead time: the HBM write stream of the
  // other waves stretches the load latency, profiles/r04_notes.md 12) and therefore ahead of this group's stores
  // (vmcnt retires in order: the samples are waited for with the stores still in flight).
  constexpr bool PIPE = FW > 1 && AT_STFT_PIPE;
  constexpr bool RUNSTORE = runstore_for<M, NR>();
  constexpr int SEG_NT = (AT_STFT_SEG_NT && M == 512 && FW > 1) ? 1 : 0;
  float2 nxt16[PIPE ? 16 : 1];

  // ---- one frame group: window, FFT, split, stores (+ mel).  `a` holds the raw samples.
  // Returns true when the loads of the next consecutive frame were issued into nxt[].
  auto frame_body = [&](float2 (&a)[16], const float* __restrict__ xr, int64_t row, int gb, int64_t s0,
                        bool want_next) __attribute__((always_inline)) -> bool {
    const int fo = min(gb * FW + fs, n_out - 1);   // output frame of this thread's slot; slots past the row's end repeat its last frame
    [[maybe_unused]] const bool live = true;
    bool have_nxt = false;
    if constexpr (PIPE) {
      // the next group of this wave's run: same row, every slot a real interior frame (wave-uniform), vector loads
      if (VEC2 && want_next && (gb + 2) * FW <= n_out) {
        const int64_t s0n = s0 + (int64_t)FW * A.hop;
        if (__all(A.pad == 0 && s0n >= 0 && s0n + N <= Ti)) {
          have_nxt = true;
          const float2* __restrict__ p = reinterpret_cast<const float2*>(xr + s0n);
#pragma unroll
          for (int q = 0; q < 16; ++q) nxt16[q] = p[t + L * q];
        }
      }
    }
    if (can_reuse) {
#pragma unroll
      for (int q = 0; q < 16; ++q) raw[q] = a[q];
    }
    {
      const float2* w2 = reinterpret_cast<const float2*>(s_win);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float2 w = w2[t + L * q];
        a[q] = make_float2(a[q].x * w.x, a[q].y * w.y);
      }
    }
    float2* __restrict__ orow = A.out + ((int64_t)row * n_out + fo) * (M + 1);

    if (AT_STFT_DEBUGMODES && A.debug == 2) {  // store-only experiment: same addresses, no transform
      if (live) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { orow[t + L * q] = a[q]; orow[M - (t + L * q)] = a[q + 8]; }
        if (t == 0) orow[M / 2] = a[0];
      }
      return false;
    }
    // ---- complex FFT of length M (Stockham, radix 16 / R2 / R3)
    pass_compute_store<16, 1, L>(a, fbuf, t, nullptr);
    wave_sync();
    if constexpr (P::R2 > 1) {
      load_points<L>(a, fbuf, t);
      wave_sync();
      float2 tw2[NB2 * P::R2];
#pragma unroll
      for (int b = 0; b < NB2; ++b) {
        const float2* rowp = reinterpret_cast<const float2*>(s_tw2 + ((t + b * L) & 15) * 36);
#pragma unroll
        for (int r = 1; r < P::R2; ++r) tw2[b * P::R2 + r] = rowp[r];
      }
      pass_compute_store<P::R2, 16, L>(a, fbuf, t, tw2);
      wave_sync();
    }
    if constexpr (P::R3 > 1) {
      load_points<L>(a, fbuf, t);
      wave_sync();
      float2 tw3[NB3 * P::R3];
#pragma unroll
      for (int b = 0; b < NB3; ++b) {
        tw3[b * P::R3 + 1] = tw3b[b];
#pragma unroll
        for (int r = 2; r < P::R3; ++r) tw3[b * P::R3 + r] = cmul(tw3[b * P::R3 + r - 1], tw3b[b]);
      }
      pass_compute_store<P::R3, 16 * P::R2, L>(a, fbuf, t, tw3);
      wave_sync();
    }

    // ---- split step: X[k], X[M-k] from Z[k], Z[M-k]; k = t + L q, q < 8
    float2 zk[8], zm[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = t + L * q;
      zk[q] = fbuf[phys<L>(k)];
      zm[q] = fbuf[phys<L>((M - k) & (M - 1))];
    }
    float2 zh = make_float2(0.f, 0.f);
    if (t == 0) zh = fbuf[phys<L>(M / 2)];
    wave_sync();  // all reads of Z done before the slab is reused for |X|

    // Issue the next frame's new loads BEFORE this frame's stores: vmcnt retires in order, so a
    // wait for loads issued after the stores would also wait for every store acknowledgement.
    if (can_reuse && want_next && gb + 1 < gpr) {
      const int64_t s0n = s0 + A.hop;
      if (A.pad == 0 && s0n >= 0 && s0n + N <= Ti) {  // wave-uniform (FW == 1)
        have_nxt = true;
        if constexpr (VEC2) {
          const float2* __restrict__ p = reinterpret_cast<const float2*>(xr + s0n);
#pragma unroll
          for (int q = 0; q < SH; ++q) nxt[q] = p[t + L * (16 - SH + q)];
        } else {
          const float* __restrict__ p = xr + s0n;
#pragma unroll
          for (int q = 0; q < SH; ++q)
            nxt[q] = make_float2(p[2 * (t + L * (16 - SH + q))], p[2 * (t + L * (16 - SH + q)) + 1]);
        }
      }
    }

    // |X| of frame slot fs: rows of 16 bins padded to MAG_ROW floats
    float* magbase = (RUNSTORE && MEL) ? s_mag + wave * (FW * MAG_ROWS * MAG_ROW) : reinterpret_cast<float*>(wbuf);
    float* magbuf = magbase + fs * (MAG_ROWS * MAG_ROW);
    if constexpr (RUNSTORE) {
      // Several frames per wave: the FW frames of the group are consecutive rows of the output, (M + 1) FW contiguous float2.
      // Every lane drops its bins into the wave's slab (frame slot fs at the FFT's own stride 17 L: conflict-free), then the
      // wave streams the slab out as 512-byte runs -- 17 store instructions whose 64 lanes write consecutive addresses,
      // instead of 17 instructions of FW segments of 8 L bytes each (n_fft 512: 0.90 -> 0.78 ms, 256: 0.52 -> 0.42, 128:
      // 0.60 -> 0.42, 1024: 1.14 -> 1.10; profiles/r05_notes.md 8).  With the mel stage the magnitudes go to a region of
      // their own (the slab is busy with X).
      float2* sx = wbuf + fs * P::SLOTS;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = t + L * q;
        float2 xa, xb;
        if (q == 0 && t == 0) {   // k == 0: DC and Nyquist (Z is computed from the halved window)
          xa = make_float2(2.f * (zk[0].x + zk[0].y), 0.f);
          xb = make_float2(2.f * (zk[0].x - zk[0].y), 0.f);
        } else {
          const float2 twp = s_twp[k];
          const float sr = zk[q].x + zm[q].x, si = zk[q].y - zm[q].y;
          const float dr = zk[q].x - zm[q].x, di = zk[q].y + zm[q].y;
          const float c = twp.x, s = -twp.y;
          const float pp = fmaf(s, dr, -c * di);
          const float qq = fmaf(s, di, c * dr);
          xa = make_float2(sr - pp, si - qq);
          xb = make_float2(sr + pp, -si - qq);
        }
        sx[k] = xa;
        sx[M - k] = xb;
        if constexpr (MEL) {
          magbuf[k + 4 * (k >> 4)] = cabs_fast(xa);
          magbuf[(M - k) + 4 * ((M - k) >> 4)] = cabs_fast(xb);
        }
      }
      if (t == 0) {
        const float2 xh = make_float2(2.f * zh.x, -2.f * zh.y);
        sx[M / 2] = xh;
        if constexpr (MEL) magbuf[M / 2 + 4 * ((M / 2) >> 4)] = cabs_fast(xh);
      }
      wave_sync();
      constexpr int TOT = FW * (M + 1);
      constexpr int NST = (TOT + 63) / 64;
      const int f0 = gb * FW;
      float2* __restrict__ obase = A.out + ((int64_t)row * n_out + f0) * (M + 1);
      if (f0 + FW <= n_out) {               // wave-uniform: every slot a frame of its own -> unconditional stores
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int idx = min(i * 64 + lane, TOT - 1);      // (only the last instruction clamps: duplicates of the last bin)
          const int f = idx / (M + 1);
          st2<AT_STFT_RUNSTORE_NT>(obase + idx, wbuf[f * P::SLOTS + (idx - f * (M + 1))]);
        }
      } else {                              // last group of a row: only the frames that exist
        const int tot = (n_out - f0) * (M + 1);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int idx = i * 64 + lane;
          const int f = idx / (M + 1);
          if (idx < tot) obase[idx] = wbuf[f * P::SLOTS + (idx - f * (M + 1))];
        }
      }
      if constexpr (!MEL) wave_sync();      // every read of the slab is done (the mel stage below ends with the same)
    } else {
    // Lane t computes the pair X[k], X[M-k] for k = t + L q.  X[k] is stored directly
    // (addresses ascend with the lane).  The partner X[M-k] DEscends with the lane, and a
    // store instruction whose lanes write descending addresses is ~10 % slower on this memory
    // system (tools/micro/wrbench), so the partners are reversed across the L lanes of the frame
    // (one ds_bpermute per component): lane t then holds bin (M - L q - L + 1) + t, ascending.
    const int rev_lane = (lane - t) + (L - 1 - t);
    float2 xbs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = t + L * q;
      float2 xa, xb;
      if (q == 0 && t == 0) {  // k == 0: DC (here) and Nyquist (kept as this lane's partner)
        xa = make_float2(2.f * (zk[0].x + zk[0].y), 0.f);   // (Z is computed from the halved window)
        xb = make_float2(2.f * (zk[0].x - zk[0].y), 0.f);
      } else {
        const float2 twp = s_twp[k];  // (cos, -sin)(2 pi k / N)
        const float sr = zk[q].x + zm[q].x, si = zk[q].y - zm[q].y;
        const float dr = zk[q].x - zm[q].x, di = zk[q].y + zm[q].y;
        const float c = twp.x, s = -twp.y;
        const float pp = fmaf(s, dr, -c * di);
        const float qq = fmaf(s, di, c * dr);
        xa = make_float2(sr - pp, si - qq);          // the 1/2 is in the window table
        xb = make_float2(sr + pp, -si - qq);
      }
      if (STORE_OK(xa.x)) st2<SEG_NT>(orow + k, xa);
      if constexpr (MEL) magbuf[k + 4 * (k >> 4)] = cabs_fast(xa);
      xbs[q] = make_float2(__shfl(xb.x, rev_lane, 64), __shfl(xb.y, rev_lane, 64));
    }
    // ascending partner segments: q = 7 .. 0
#pragma unroll
    for (int q = 7; q >= 0; --q) {
      const int kb = (M - L * q - L + 1) + t;   // lane L-1 of q == 0 holds the Nyquist bin M
      if (STORE_OK(xbs[q].x)) st2<SEG_NT>(orow + kb, xbs[q]);
      if constexpr (MEL) magbuf[kb + 4 * (kb >> 4)] = cabs_fast(xbs[q]);
    }
    if (t == 0) {  // k == M/2: X = conj(Z[M/2])
      const float2 xh = make_float2(2.f * zh.x, -2.f * zh.y);
      if (STORE_OK(xh.x)) orow[M / 2] = xh;
      if constexpr (MEL) magbuf[M / 2 + 4 * ((M / 2) >> 4)] = cabs_fast(xh);
    }
    }
    if constexpr (MEL) {
      // The last row holds only the Nyquist bin; a unit on it multiplies its other 15 columns by
      // zero weights, and those columns alias FFT-slab slots that may never have been written
      // (0 * garbage = NaN).  Zero them.
      for (int c = 1 + t; c < 16; c += L) magbuf[(M / 16) * MAG_ROW + c] = 0.f;
    }

    if constexpr (MEL) {
      wave_sync();
#pragma unroll AT_STFT_MELUNROLL
      for (int fsl = 0; fsl < FW; ++fsl) {
        const int fo2 = gb * FW + fsl;
        if (fo2 >= n_out) break;  // wave-uniform
        const float* mg = magbase + fsl * (MAG_ROWS * MAG_ROW);
        float* mrow = A.mel + ((int64_t)row * n_out + fo2) * A.n_mels;
        // two rounds at a time: enough independent work to cover the LDS latency, while the
        // scheduling barrier keeps the other rounds' 32 loaded registers each from piling up
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += 2) {
          float acc[2];
          int u_fl[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int r = r0 + j;
            int u_off;
            if constexpr (MELDESC) {
              u_off = mu_off[r];
              u_fl[j] = mu_fl[r];
            } else {
              const int2 ui = s_uinfo[r * 64 + lane];
              u_off = ui.x;
              u_fl[j] = ui.y;
            }
            const float4* mq = reinterpret_cast<const float4*>(mg + u_off);
            const float4* wq = reinterpret_cast<const float4*>(s_melw + (r * 64 + lane) * MELW_ROW);
            float v = 0.f;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
              float4 w;
              if constexpr (MELREG) w = mu_w[r][i4]; else w = wq[i4];
              const float4 m = mq[i4];
              v = fmaf(w.x, m.x, v);
              v = fmaf(w.y, m.y, v);
              v = fmaf(w.z, m.z, v);
              v = fmaf(w.w, m.w, v);
            }
            acc[j] = v;
          }
          // segmented sums over the adjacent lanes (inside one 16-lane row) that hold one band:
          // DPP row_shl:n hands lane i the value of lane i+n
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float sh;  // the DPP read must run in every lane: shuffle first, select afterwards
            sh = dpp_row_shl<1>(acc[j]); acc[j] += (u_fl[j] & 1) ? sh : 0.f;
            sh = dpp_row_shl<2>(acc[j]); acc[j] += (u_fl[j] & 2) ? sh : 0.f;
            sh = dpp_row_shl<4>(acc[j]); acc[j] += (u_fl[j] & 4) ? sh : 0.f;
            sh = dpp_row_shl<8>(acc[j]); acc[j] += (u_fl[j] & 8) ? sh : 0.f;
            if (u_fl[j] & 16) mrow[u_fl[j] >> 8] = acc[j];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      wave_sync();
    }
    return have_nxt;
  };

  for (int64_t gbase = w0; gbase < g_hi; gbase += Wtot * RUN) {
    const int64_t g_end = min(gbase + RUN, g_hi);
    int64_t g = gbase;
    while (g < g_end) {
      // ---- first frame of a stretch: all 16 loads (or the edge path)
      const int64_t row = g / gpr;
      int gb = (int)(g - row * gpr);
      const float* __restrict__ xr = A.x + row * A.T;
      const int fo = min(gb * FW + fs, n_out - 1);          // (slots past the row's last frame repeat it)
      const bool live = true;
      int64_t s0 = ((int64_t)fo + A.frame_lo) * A.hop - M;  // first sample, outer-padded coords
      const bool interior = A.pad == 0 && s0 >= 0 && s0 + N <= Ti;
      float2 a[16];
      if (__all(interior)) {
        if (!live) {
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = make_float2(0.f, 0.f);
        } else if constexpr (VEC2) {
          const float2* __restrict__ p = reinterpret_cast<const float2*>(xr + s0);
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = p[t + L * q];
        } else {
          const float* __restrict__ p = xr + s0;
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = make_float2(p[2 * (t + L * q)], p[2 * (t + L * q) + 1]);
        }
      } else {
        // edge frames (padding by index math): rolled loop through the wave's LDS slab, which is
        // free here; keeps the register arrays statically indexed
#pragma unroll 1
        for (int q = 0; q < 16; ++q) {
          const int n = t + L * q;
          float2 v = make_float2(0.f, 0.f);
          if (live)
            v = make_float2(fetch_padded(xr, s0 + 2 * n, A.T, A.T2, A.pad, A.pad_mode),
                            fetch_padded(xr, s0 + 2 * n + 1, A.T, A.T2, A.pad, A.pad_mode));
          fbuf[n] = v;
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = fbuf[t + L * q];
        wave_sync();
      }
      bool more = frame_body(a, xr, row, gb, s0, g + 1 < g_end);
      ++g;
      // ---- steady state: consecutive frames of the same row.  This loop is ONLY entered with
      // "SH loads, then the stores of one frame" in flight on every path, so the compiler's
      // s_waitcnt for nxt[] tolerates the stores (vmcnt(17+)) instead of draining them.
      while (more) {
        ++gb;
        float2 b[16];
        if constexpr (PIPE) {
          s0 += (int64_t)FW * A.hop;
#pragma unroll
          for (int q = 0; q < 16; ++q) b[q] = nxt16[q];
        } else {
          s0 += A.hop;
#pragma unroll
          for (int q = 0; q < 16 - SH; ++q) b[q] = raw[q + SH];
#pragma unroll
          for (int q = 0; q < SH; ++q) b[16 - SH + q] = nxt[q];
        }
        more = frame_body(b, xr, row, gb, s0, g + 1 < g_end);
        ++g;
      }
    }
  }
}


// =============================================================================================
// n_fft = 2048 / hop = 512 specialisation (the reference's default parameters at 44.1 kHz,
// audio_signal.py:1066-1070): same algorithm as stft_mel_kernel<1024, ...> with three changes
// aimed at the two busiest units of that kernel (LDS array ~55 %, VALU ~41 % of the launch):
//  * PAIRED LAST PASS.  The final radix-4 pass gives every thread both members of each
//    (k, M - k) pair: thread t runs butterflies t, 256 - t, 64 + t, 192 - t (thread 0: 0, 128, 64,
//    192), so the real-FFT split step runs on registers -- no slab write of Z (16 ds_write_b64)
//    and no re-read of Z[k], Z[M-k] (33 ds_read_b64, half of them 2-way bank conflicts).  The
//    partner outputs X[M-k] descend with the lane; they are handed to lane (64 - t) mod 64 with
//    one ds_bpermute per dword so that every store instruction writes 512 contiguous bytes with
//    ascending lanes (tools/emulate_stft_v2.py is the lane-level model of this index algebra);
//  * NO REGISTER SHUFFLING.  With hop = n_fft/4 consecutive frames share 12 of 16 sample
//    registers.  The sample registers stay where they are and the frame PHASE (0..3) rotates the
//    logical view (logical q lives in raw[(q + 4 phase) & 15]); the 4 new loads of the next frame
//    land directly in the 4 registers the current frame consumed first (the generic kernel spends
//    ~145 v_mov per frame on the shift);
//  * window and pass-2 twiddles are read as ds_read_b128 rows (the stride-64 float2 reads were
//    merged by the compiler into half-rate ds_read2st64_b64).
// POL = cache policy of the streaming traffic (AT_STFT_FLAGS, measurement knob; 0 is shipped):
//   bit 0: spectrum stores `nt`;  bit 1: mel stores `nt`;  bit 2: the sample loads `nt`;  bit 3: raised wave priority (s_setprio 3) while
//   the loads and stores of a frame are issued.  (`sc1` write-through stores were measured 25 % slower.)
template <int POL>
__device__ __forceinline__ void st2(float2* p, float2 v) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  if constexpr (POL & 1) {
    const v2f u = {v.x, v.y};
    __builtin_nontemporal_store(u, reinterpret_cast<v2f*>(p));
  } else {
    *p = v;
  }
}
template <int POL>
__device__ __forceinline__ float2 ld2(const float2* p) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  if constexpr (POL & 4) {
    const v2f u = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p));
    return make_float2(u.x, u.y);
  } else {
    return *p;
  }
}

// VAR bit 0: STATIC STORE COUNT.  The 4 sample loads of the next frame are issued before this frame's stores,
//   and vmcnt retires in order: the wait for them at the top of the next frame is  vmcnt(#stores behind
//   them).  The compiler can only count stores that are issued on every path; the Nyquist store under
//   `if (lane == 0)` and the four band stores under `if (lane heads a band)` sit behind exec-mask branches,
//   so it assumed 16 stores where 17 (21 with mel) are in flight, and every frame began by waiting for the
//   acknowledgement of the first 1 (5) stores of the previous frame (ISA: s_waitcnt vmcnt(19..16) at the
//   loop head, profiles/r03_notes.md).  With VAR & 1 the Nyquist bin is stored by all lanes (same
//   address, lane 0's value) and the band sums are gathered in the wave's LDS slab and stored by two
//   unconditional, coalesced instructions: every store is counted, the loop head waits for the loads only.
// VAR bit 1: FLOOR.  The measurement twin of the kernel (at_stft_mel_floor_f32, round 4): the same persistent grid, runs,
//   XCD spans, row pitch, load / store instructions, counts, order and cache policy -- and no transform: a frame's stores
//   carry its sample registers.  Its duration on a box is what this traffic pattern costs there with zero compute;
//   bench.py prints it next to the kernel's (roofline.floor_ms).  Output contents are meaningless.
template <int NR /* mel rounds of 64 units; 0 = no mel */, int POL = 0, int VAR = 0>
__global__ __launch_bounds__(256, 2) void stft_mel_kernel_v2(const StftArgs A) {
  constexpr bool MEL = NR > 0;
  constexpr bool STATIC_STORES = (VAR & 1) != 0;
  constexpr bool FLOOR = (VAR & 2) != 0;
  // VAR bits 2-3: PIPE.  The unit rounds of frame f's mel run at the top of frame f + 1, between the register-only part of its
  // first pass (window, radix-16 butterfly) and that pass's slab stores: the mel stage's LDS round trips and DPP chains (a serial
  // chain of ~1000 cycles per frame that two waves per SIMD did not hide) share their issue slots with ~200 independent VALU
  // instructions.  The magnitudes stay in the slab until then (pass 1 overwrites it only after the last unit read); the band sums
  // are gathered in a per-wave region of their own and leave with the exchange reads of pass 1.  A stretch drains its last frame.
  constexpr int PIPE = (MEL && STATIC_STORES && !FLOOR) ? ((VAR >> 2) & 3) : 0;
  constexpr bool HX = !FLOOR && (VAR & 16) != 0;      // VAR bit 4: half of the pass 2 -> 3 exchange in registers (see pass 3)
  constexpr int MELOUT_OFF = 1312;   // floats: band sums of the frame, behind the 65 x 20 magnitude rows of the slab
  constexpr bool PRIO = (POL & 8) != 0;
  constexpr int M = 1024, L = 64, N = 2048, NW = 4, SH = 4;
  constexpr int WROW = 36;  // floats per window row (32 used): conflict-free ds_read_b128
  constexpr int TROW = 20;  // floats per split-twiddle row (16 used): conflict-free ds_read_b128
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float* s_winr = lds_f + NW * 2 * WAVE_LDS_SLOTS;                       // [64 threads][16 q] float2, halved
  float* s_twp = s_winr + 64 * WROW;              // [64 t] rows of 20 floats: split twiddles, pair A r=0..3, pair B r=0..3
  float* s_tw2 = s_twp + 64 * TROW;                // 16 rows x 36: pass-2 twiddles r = 1..15 of row (j mod 16)
  float* s_melw = s_tw2 + 16 * 36;
  int2* s_uinfo = reinterpret_cast<int2*>(s_melw + A.n_units * MELW_ROW);

  for (int i = threadIdx.x; i < N; i += NW * 64) {
    const int n2 = i >> 1;
    s_winr[(n2 & 63) * WROW + 2 * (n2 >> 6) + (i & 1)] = 0.5f * A.window[i];
  }
  for (int i = threadIdx.x; i < 256; i += NW * 64) {
    const int r = i >> 6, tt = i & 63;
    // thread 0 of pair A runs the self-paired butterflies 0 and 128: its slots use k = 128, 256, 384
    const int kA = tt == 0 ? (r < 3 ? 128 * (r + 1) : 0) : tt + 256 * r;
    reinterpret_cast<float2*>(s_twp + tt * TROW)[r] = A.tw[kA];
    reinterpret_cast<float2*>(s_twp + tt * TROW)[4 + r] = A.tw[64 + tt + 256 * r];
  }
  for (int i = threadIdx.x; i < 16 * 16; i += NW * 64) {
    const int jj = i / 16, r = i % 16;   // column r - 1 holds w_256^(jj r); the last column is padding
    reinterpret_cast<float2*>(s_tw2 + jj * 36)[(r + 15) & 15] = r == 0 ? make_float2(0.f, 0.f) : A.tw[r * jj * (N / 256)];
  }
  if constexpr (MEL) {
    for (int i = threadIdx.x; i < A.n_units * 16; i += NW * 64)
      s_melw[(i >> 4) * MELW_ROW + (i & 15)] = A.unit_w[i];
    for (int i = threadIdx.x; i < A.n_units; i += NW * 64) {
      const int info = A.unit_info[2 * i], fl = A.unit_info[2 * i + 1];
      s_uinfo[i] = make_int2((info & 0xffff) * MAG_ROW, (fl & 0xff) | (((info >> 16) & 0xffff) << 8));
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane;
  const bool t0 = t == 0;
  float2* fbuf = reinterpret_cast<float2*>(lds_f) + wave * WAVE_LDS_SLOTS;
  float* magbuf = reinterpret_cast<float*>(fbuf);

  // butterflies of the paired last pass and their base twiddles w_1024^j
  const int jb[4] = {t, t0 ? 128 : 256 - t, 64 + t, 192 - t};
  int pj[4];
  float2 twb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    pj[b] = phys<L, true>(jb[b]);
    twb[b] = A.tw[2 * jb[b]];
  }
  const int src_lane4 = ((64 - lane) & 63) << 2;   // ds_bpermute byte index of the partner lane
  // this lane's mel units (one per round): magnitude row, the four tree-step bits as bytes, band to store
  int m_off[MEL ? NR : 1], m_fl[MEL ? NR : 1], m_st[MEL ? NR : 1];
  if constexpr (MEL) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int2 ui = s_uinfo[r * 64 + lane];
      const int fl = ui.y;
      m_off[r] = ui.x;
      m_fl[r] = (fl & 1) | (((fl >> 1) & 1) << 8) | (((fl >> 2) & 1) << 16) | (((fl >> 3) & 1) << 24);
      m_st[r] = (fl & 16) ? (fl >> 8) : (PIPE != 0 ? 128 + (lane & 15) : -1);   // PIPE: lanes that head no band write a dump slot
    }
  }
  // STATIC_STORES: the bands this lane stores (lane, 64 + lane, clamped to the last band: duplicates of it are harmless)
  const int gb0 = (MEL && STATIC_STORES) ? min(lane, A.n_mels - 1) : 0;
  const int gb1 = (MEL && STATIC_STORES) ? min(64 + lane, A.n_mels - 1) : 0;

  const int Ti = (int)A.T;
  const int n_out = (int)A.n_out;
  const int gpr = A.groups_per_row;
  const int n_x = (int)gridDim.x < A.n_xcd ? (int)gridDim.x : A.n_xcd;
  const int xcd = blockIdx.x % n_x;
  const int lblk = blockIdx.x / n_x;
  const int nblk_x = ((int)gridDim.x - xcd + n_x - 1) / n_x;
#if AT_DEV_KNOBS
  // flags bits 14-16 (xor mask) and 17-18 (multiplier 1, 3, 5, 7): which span an XCD takes (8 spans only)
  const int reg = n_x == 8 ? (((xcd * (2 * ((A.flags >> 17) & 3) + 1)) ^ ((A.flags >> 14) & 7)) & 7) : xcd;
#else
  const int reg = xcd;
#endif
  const int64_t g_lo = A.total_groups * reg / n_x;
  const int64_t g_hi = A.total_groups * (reg + 1) / n_x;
  const int RUN = A.run;
  const int64_t Wtot = (int64_t)nblk_x * NW;
  const int64_t w0 = g_lo + ((int64_t)lblk * NW + wave) * RUN;
  // De-phase the 8 resident waves of a CU.  They all start together and run identical work, so
  // without a stagger the chip alternates between "everyone transforms" and "everyone stores" and
  // the memory system idles part of the time; slot s starts s * stagger * 64 cycles late.
  if (A.stagger > 0) {
    const int slot = (wave + NW * (lblk & 1)) & 7;
    for (int i = 0; i < slot * A.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  if ((A.flags & 16) && (lblk & 1)) __builtin_amdgcn_s_setprio(2);   // measurement knob: the two waves of a SIMD at different priority

  float2 raw[16];   // un-windowed samples; logical q of a phase-p frame is raw[(q + 4 p) & 15]
  float2 nxt[SH];   // the 4 new loads of the NEXT frame, issued ahead of this frame's stores

  // ---- mel stage: unit dot products as packed FMAs on the float4 rows as they come out of LDS (two independent
  // accumulator pairs: dependency depth 6 instead of 16); the segmented sums take their participation bits as 0.0 / 1.0
  // factors, one FMA per step.  Software pipeline over the rounds with two operand buffers: the reads of round r + 2 are
  // issued as soon as round r's products are done, and the (serial) DPP chain of round r - 1 shares its issue slots with
  // the products of round r.
  typedef float v2f __attribute__((ext_vector_type(2)));
  struct MelOps { float4 wv[2][4], mv[2][4]; };
  // band sums of a frame: in the slab behind the magnitude rows, or (PIPE) in a region of the wave's own behind the tables
  float* const melout = PIPE ? reinterpret_cast<float*>(s_uinfo + A.n_units) + wave * 144 : magbuf + MELOUT_OFF;
  auto load_unit = [&](MelOps& o, int r, int b) __attribute__((always_inline)) {
    const float4* mq = reinterpret_cast<const float4*>(magbuf + m_off[r]);
    const float4* wq = reinterpret_cast<const float4*>(s_melw + (r * 64 + lane) * MELW_ROW);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) { o.wv[b][i4] = wq[i4]; o.mv[b][i4] = mq[i4]; }
  };
  auto dot_unit = [&](const MelOps& o, int b) __attribute__((always_inline)) -> float {
    v2f e = v2f{o.wv[b][0].x, o.wv[b][0].y} * v2f{o.mv[b][0].x, o.mv[b][0].y};
    v2f od = v2f{o.wv[b][0].z, o.wv[b][0].w} * v2f{o.mv[b][0].z, o.mv[b][0].w};
#pragma unroll
    for (int i4 = 1; i4 < 4; ++i4) {
      e = __builtin_elementwise_fma(v2f{o.wv[b][i4].x, o.wv[b][i4].y}, v2f{o.mv[b][i4].x, o.mv[b][i4].y}, e);
      od = __builtin_elementwise_fma(v2f{o.wv[b][i4].z, o.wv[b][i4].w}, v2f{o.mv[b][i4].z, o.mv[b][i4].w}, od);
    }
    const v2f sum = e + od;
    return sum.x + sum.y;
  };
  auto reduce_store = [&](float acc, int r, float* mrow) __attribute__((always_inline)) {
    const unsigned f = (unsigned)m_fl[r];
    float sh;
    sh = dpp_row_shl<1>(acc); acc = fmaf(sh, (float)(f & 0xffu), acc);
    sh = dpp_row_shl<2>(acc); acc = fmaf(sh, (float)((f >> 8) & 0xffu), acc);
    sh = dpp_row_shl<4>(acc); acc = fmaf(sh, (float)((f >> 16) & 0xffu), acc);
    sh = dpp_row_shl<8>(acc); acc = fmaf(sh, (float)(f >> 24), acc);
    if constexpr (PIPE != 0) {
      melout[m_st[r]] = acc;          // every lane (m_st: band, or a dump slot behind the bands): no exec-mask branch cuts the slot
    } else if constexpr (STATIC_STORES) {
      if (m_st[r] >= 0) melout[m_st[r]] = acc;
    } else {
      if (m_st[r] >= 0) at::stg<(POL & 2) != 0>(mrow + m_st[r], acc);
    }
  };
  // all rounds of one frame (the magnitudes are in the slab); SCHED: hold the round pipeline in place with scheduling barriers
  auto mel_rounds = [&](MelOps& o, float* mrow, auto sched_c) __attribute__((always_inline)) {
    constexpr bool SCHED = decltype(sched_c)::value;
    float acc_prev = dot_unit(o, 0);
    if constexpr (SCHED) __builtin_amdgcn_sched_barrier(0);
    if constexpr (NR > 2) load_unit(o, 2, 0);
#pragma unroll
    for (int r = 1; r < NR; ++r) {
      const float acc_cur = dot_unit(o, r & 1);
      reduce_store(acc_prev, r - 1, mrow);
      if constexpr (SCHED) __builtin_amdgcn_sched_barrier(0);
      if (r + 2 < NR) load_unit(o, r + 2, r & 1);
      acc_prev = acc_cur;
    }
    reduce_store(acc_prev, NR - 1, mrow);
  };
  // the gathered band sums leave as two unconditional, coalesced stores (STATIC_STORES)
  auto melout_store = [&](float* mrow) __attribute__((always_inline)) {
    const float b0 = melout[gb0], b1 = melout[gb1];
    at::stg<(POL & 2) != 0>(mrow + gb0, b0);
    at::stg<(POL & 2) != 0>(mrow + gb1, b1);
  };
  // PIPE: the mel of a stretch's last frame (nothing follows it to hide behind)
  auto mel_drain = [&](int64_t row, int fo) __attribute__((always_inline)) {
    float* mrow = A.mel + ((int64_t)row * n_out + fo) * A.n_mels;
    wave_sync();
    MelOps o;
    load_unit(o, 0, 0);
    load_unit(o, 1, 1);
    mel_rounds(o, mrow, std::true_type{});
    wave_sync();
    melout_store(mrow);
    wave_sync();
  };

  auto split = [&](float2 zk, float2 zm, float2 twp, float2& xa, float2& xb) __attribute__((always_inline)) {
    const float sr = zk.x + zm.x, si = zk.y - zm.y;
    const float dr = zk.x - zm.x, di = zk.y + zm.y;
    const float c = twp.x, s = -twp.y;
    const float pp = fmaf(s, dr, -c * di);
    const float qq = fmaf(s, di, c * dr);
    xa = make_float2(sr - pp, si - qq);      // X[k]      (the 1/2 is in the window table)
    xb = make_float2(sr + pp, -si - qq);     // X[M - k]
  };
  auto to_partner = [&](float2 v) __attribute__((always_inline)) -> float2 {
    return make_float2(
        __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, v.x))),
        __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, v.y))));
  };

  // one frame of phase p.  Returns true when the 4 new loads of the next frame were issued.
  auto frame_body = [&](auto phase_c, auto steady_c, const float* __restrict__ xr, int64_t row, int fo, int64_t s0,
                        bool want_next) __attribute__((always_inline)) -> bool {
    constexpr int P = decltype(phase_c)::value;
    if constexpr (decltype(steady_c)::value) {
      // the 4 samples loaded during the previous frame become logical q = 12..15 of this one
#pragma unroll
      for (int i = 0; i < SH; ++i) raw[(12 + i + 4 * P) & 15] = nxt[i];
    }
    if constexpr (FLOOR) {
      const int64_t s0n = s0 + A.hop;
      const bool have_nxt = want_next && fo + 1 < gpr && s0n >= 0 && s0n + N <= Ti;
#if AT_DEV_KNOBS
      // development twins of the twin (tools/regime.py): flags bit 6 = no sample loads, bit 7 = no stores (the loaded
      // samples are folded into a value that is stored only if it equals a constant it never takes)
      if (A.flags & 64) {
#pragma unroll
        for (int i = 0; i < SH; ++i) nxt[i] = make_float2(1.f, 2.f);
      } else
#endif
      {
        const float2* __restrict__ p2 = reinterpret_cast<const float2*>(xr + (have_nxt ? s0n : 0)) + t + L * (16 - SH);
#pragma unroll
        for (int i = 0; i < SH; ++i) nxt[i] = ld2<POL>(p2 + L * i);
      }
      float2* __restrict__ orow = A.out + ((int64_t)row * n_out + fo) * (M + 1);
#if AT_DEV_KNOBS
      if (A.flags & 128) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += raw[q].x * raw[q].y;
        if (acc == 12345.678f) st2<POL>(orow + t, make_float2(acc, 0.f));
        return have_nxt;
      }
#endif
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k0 = 256 * m + t;
        st2<POL>(orow + k0, raw[(m + 4 * P) & 15]);
        st2<POL>(orow + k0 + 64, raw[(4 + m + 4 * P) & 15]);
        st2<POL>(orow + k0 + 128, raw[(8 + m + 4 * P) & 15]);
        st2<POL>(orow + k0 + 192, raw[(12 + m + 4 * P) & 15]);
      }
      const float nyq0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, raw[(4 * P) & 15].x), 0));
      st2<POL>(orow + M, make_float2(nyq0, 0.f));
      if constexpr (MEL) {
        float* mrow = A.mel + ((int64_t)row * n_out + fo) * A.n_mels;
        at::stg<(POL & 2) != 0>(mrow + gb0, raw[(1 + 4 * P) & 15].x);
        at::stg<(POL & 2) != 0>(mrow + gb1, raw[(2 + 4 * P) & 15].y);
      }
      return have_nxt;
    } else {
    // PIPE: this frame carries the unit rounds of its predecessor (every frame of a stretch but the first)
    constexpr bool PIPE_NOW = PIPE != 0 && decltype(steady_c)::value;
    MelOps mo;
    float* mrow_prev = nullptr;
    if constexpr (PIPE_NOW) {
      mrow_prev = A.mel + ((int64_t)row * n_out + (fo - 1)) * A.n_mels;
      wave_sync();
      load_unit(mo, 0, 0);
      if constexpr (PIPE != 3) load_unit(mo, 1, 1);
    }
    float2 a[16];
    {
      const float4* wr = reinterpret_cast<const float4*>(s_winr + t * WROW);
      float4 w4[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w4[i] = wr[i];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float2 r_ = raw[(q + 4 * P) & 15];
        const float4 w_ = w4[q >> 1];
        a[q] = (q & 1) ? make_float2(r_.x * w_.z, r_.y * w_.w) : make_float2(r_.x * w_.x, r_.y * w_.y);
      }
    }
    // ---- passes 1 and 2 (radix 16, 16) through the slab
    if constexpr (PIPE_NOW) {
      // pass 1's butterfly in registers, the previous frame's unit rounds beside it, then the slab stores
      if constexpr (PIPE == 1) {
        Dft<16>::run(a);
        mel_rounds(mo, mrow_prev, std::false_type{});
      } else {
        // SLOTS: the radix-16 butterfly cut into five pieces (as Dft<16>::run: four 4-point transforms on the columns in two
        // pieces, the w_16 twiddles, four 4-point transforms on the rows in two pieces), one per slot, each slot also holding one
        // step of the round pipeline: the products of round s, the segmented sum of round s - 1, the LDS reads of round s + 2.
        // Scheduling barriers keep the slots apart; inside a slot the compiler orders freely.
        float2 C[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) C[c][r] = a[c + 4 * r];
        float acc[NR + 1];
        auto mel_step = [&](auto s_c) __attribute__((always_inline)) {
          constexpr int S = decltype(s_c)::value;
          if constexpr ((AT_STFT_ABL & 2) != 0) {
            if constexpr (S == 0) melout[m_st[0]] = mo.mv[0][0].x;
          } else if constexpr (PIPE == 3) {          // one operand buffer: the reads of round s + 1 follow the products of round s
            if constexpr (S < NR) acc[S] = dot_unit(mo, 0);
            if constexpr (S + 1 < NR) load_unit(mo, S + 1, 0);
            if constexpr (S >= 1 && S <= NR) reduce_store(acc[S - 1], S - 1, mrow_prev);
          } else {
            if constexpr (S < NR) acc[S] = dot_unit(mo, S & 1);
            if constexpr (S >= 1 && S <= NR) reduce_store(acc[S - 1], S - 1, mrow_prev);
            if constexpr (S + 2 < NR) load_unit(mo, S + 2, S & 1);
          }
        };
        auto dft_piece = [&](auto p_c) __attribute__((always_inline)) {
          constexpr int Pc = decltype(p_c)::value;
          if constexpr (Pc == 0) { dft4(C[0][0], C[0][1], C[0][2], C[0][3]); dft4(C[1][0], C[1][1], C[1][2], C[1][3]); }
          if constexpr (Pc == 1) { dft4(C[2][0], C[2][1], C[2][2], C[2][3]); dft4(C[3][0], C[3][1], C[3][2], C[3][3]); }
          if constexpr (Pc == 2) {
            C[1][1] = mul_w16<1>(C[1][1]); C[1][2] = mul_w16<2>(C[1][2]); C[1][3] = mul_w16<3>(C[1][3]);
            C[2][1] = mul_w16<2>(C[2][1]); C[2][2] = mul_w16<4>(C[2][2]); C[2][3] = mul_w16<6>(C[2][3]);
            C[3][1] = mul_w16<3>(C[3][1]); C[3][2] = mul_w16<6>(C[3][2]); C[3][3] = mul_w16<9>(C[3][3]);
          }
          if constexpr (Pc == 3 || Pc == 4) {
#pragma unroll
            for (int k1 = 2 * (Pc - 3); k1 < 2 * (Pc - 3) + 2; ++k1) {
              float2 b0 = C[0][k1], b1 = C[1][k1], b2 = C[2][k1], b3 = C[3][k1];
              dft4(b0, b1, b2, b3);
              a[k1] = b0; a[k1 + 4] = b1; a[k1 + 8] = b2; a[k1 + 12] = b3;
            }
          }
        };
        dft_piece(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mel_step(std::integral_constant<int, 0>{}); dft_piece(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mel_step(std::integral_constant<int, 1>{}); dft_piece(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        mel_step(std::integral_constant<int, 2>{}); dft_piece(std::integral_constant<int, 3>{});
        __builtin_amdgcn_sched_barrier(0);
        mel_step(std::integral_constant<int, 3>{}); dft_piece(std::integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        mel_step(std::integral_constant<int, 4>{});
        if constexpr (NR > 4) {
          __builtin_amdgcn_sched_barrier(0);
          mel_step(std::integral_constant<int, 5>{});
          __builtin_amdgcn_sched_barrier(0);
          mel_step(std::integral_constant<int, 6>{});
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) fbuf[phys<L, true>(16 * t + r)] = a[r];
    } else {
      pass_compute_store<16, 1, L, true>(a, fbuf, t, nullptr);
    }
    wave_sync();
    load_points<L, true>(a, fbuf, t);
    float pb0 = 0.f, pb1 = 0.f;
    if constexpr (PIPE_NOW) { pb0 = melout[gb0]; pb1 = melout[gb1]; }
    wave_sync();
    if constexpr (PIPE_NOW) {
      at::stg<(POL & 2) != 0>(mrow_prev + gb0, pb0);
      at::stg<(POL & 2) != 0>(mrow_prev + gb1, pb1);
    }
    {
      float2 tw2[16];
      const float4* rp = reinterpret_cast<const float4*>(s_tw2 + (t & 15) * 36);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = rp[i];
        tw2[2 * i + 1] = make_float2(v.x, v.y);
        if (i < 7) tw2[2 * i + 2] = make_float2(v.z, v.w);
      }
      if constexpr (!HX) {
        pass_compute_store<16, 16, L, true>(a, fbuf, t, tw2);
      } else {
        // pass 2's butterfly; only the outputs the DESCENDING butterflies of pass 3 read (r = 8 .. 15: points j >= 128 of every
        // 256-block) go through the slab -- the ascending ones stay in registers (below)
#pragma unroll
        for (int r = 1; r < 16; ++r) a[r] = cmul(a[r], tw2[r]);
        Dft<16>::run(a);
        const int o0 = (t >> 4) * 256 + (t & 15);
#pragma unroll
        for (int r = 8; r < 16; ++r) fbuf[phys<L, true>(o0 + 16 * r)] = a[r];
      }
    }
    // ---- pass 3 (radix 4), paired butterflies, results stay in registers
    float2 Z[4][4];
    if constexpr (HX) {
      // HALF EXCHANGE (round 6).  After pass 2 lane 16 a + b holds, in register r, point 256 a + 16 r + b.  Butterfly j of pass 3
      // reads points j + 256 q: register j >> 4 of the four lanes 16 q + (j & 15).  For the ascending butterflies of the paired
      // pass (j = t and 64 + t) that is a 4 x 4 transpose between the wave's four 16-lane rows and the registers 0 .. 3 (4 .. 7):
      // two v_permlane16_swap + two v_permlane32_swap per dword quadruple (gfx950; tools/micro/permlane_test.hip prints their
      // semantics) instead of 8 ds_write_b64 + 8 ds_read_b64.  The descending butterflies (256 - t, 192 - t) pair lane b with
      // lane 16 - b -- and lanes 0, 16, 32, 48 with a register index one higher --: they keep the slab.
      // (the two results are read into plain unsigned variables first: `__builtin_bit_cast(float, r_[1])` on the builtin's vector
      //  result compiled to element 0 with ROCm 7.2's clang -- found in the ISA: half of the swaps were dead)
      auto swap16 = [](float& u, float& v) __attribute__((always_inline)) {
        auto r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(v), false, false);
        const unsigned x0 = r_[0], x1 = r_[1];
        u = __uint_as_float(x0); v = __uint_as_float(x1);
      };
      auto swap32 = [](float& u, float& v) __attribute__((always_inline)) {
        auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(v), false, false);
        const unsigned x0 = r_[0], x1 = r_[1];
        u = __uint_as_float(x0); v = __uint_as_float(x1);
      };
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float2 &r0 = a[4 * g], &r1 = a[4 * g + 1], &r2 = a[4 * g + 2], &r3 = a[4 * g + 3];
        swap16(r0.x, r1.x); swap16(r0.y, r1.y); swap16(r2.x, r3.x); swap16(r2.y, r3.y);
        swap32(r0.x, r2.x); swap32(r0.y, r2.y); swap32(r1.x, r3.x); swap32(r1.y, r3.y);
#pragma unroll
        for (int q = 0; q < 4; ++q) Z[2 * g][q] = a[4 * g + q];      // g = 0: butterfly t, g = 1: butterfly 64 + t
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) { Z[1][r] = fbuf[pj[1] + 257 * r]; Z[3][r] = fbuf[pj[3] + 257 * r]; }
    } else {
      wave_sync();
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[b][r] = fbuf[pj[b] + 257 * r];
    }
    wave_sync();   // slab free: it is reused for |X| below
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float2 w1 = twb[b], w2 = cmul(w1, w1), w3 = cmul(w2, w1);
      Z[b][1] = cmul(Z[b][1], w1);
      Z[b][2] = cmul(Z[b][2], w2);
      Z[b][3] = cmul(Z[b][3], w3);
      dft4(Z[b][0], Z[b][1], Z[b][2], Z[b][3]);
    }

    // ---- next frame's 4 new loads, BEFORE this frame's stores (vmcnt retires in order)
    // Issued UNCONDITIONALLY (from the row start when there is no next interior frame; T >= N is
    // a launch condition): a conditional load leaves a phi on nxt[] that the register allocator
    // resolves with copies right behind the loads, i.e. with a wait for HBM in the middle of
    // the frame.
    const int64_t s0n = s0 + A.hop;
    const bool have_nxt = want_next && fo + 1 < gpr && s0n >= 0 && s0n + N <= Ti;
    {
      const float2* __restrict__ p2 = reinterpret_cast<const float2*>(xr + (have_nxt ? s0n : 0)) + t + L * (16 - SH);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int i = 0; i < SH; ++i) nxt[i] = ld2<POL>(p2 + L * i);
    }

    // ---- split step on registers
    float2 xaA[4], xbA[4], xaB[4], xbB[4];
    float2 twp[8];
    {
      const float4* tp = reinterpret_cast<const float4*>(s_twp + t * TROW);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = tp[i];
        twp[2 * i] = make_float2(v.x, v.y);
        twp[2 * i + 1] = make_float2(v.z, v.w);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float2 zk = Z[0][r], zm = Z[1][3 - r];
      if constexpr ((AT_STFT_ABL & 8) == 0) {
      if (r == 0) zk = t0 ? Z[1][0] : zk;                 // thread 0: (Z[128], Z[896])
      if (r == 1) zm = t0 ? Z[0][3] : zm;                 // thread 0: (Z[256], Z[768])
      if (r == 2) { zk = t0 ? Z[1][1] : zk; zm = t0 ? Z[1][2] : zm; }   // thread 0: (Z[384], Z[640])
      }
      split(zk, zm, twp[r], xaA[r], xbA[r]);
      split(Z[2][r], Z[3][3 - r], twp[4 + r], xaB[r], xbB[r]);
    }
    const float2 z00 = Z[0][0], z02 = Z[0][2];
    const float2 dc = make_float2(2.f * (z00.x + z00.y), 0.f);
    const float2 nyq = make_float2(2.f * (z00.x - z00.y), 0.f);
    const float2 x512 = make_float2(2.f * z02.x, -2.f * z02.y);
    float2 ascA[4];
    if constexpr ((AT_STFT_ABL & 8) == 0) {
    ascA[0] = t0 ? dc : xaA[0];
    ascA[1] = xaA[1];
    ascA[2] = t0 ? x512 : xaA[2];
    ascA[3] = t0 ? xbA[1] : xaA[3];
    } else { ascA[0] = xaA[0]; ascA[1] = xaA[1]; ascA[2] = xaA[2]; ascA[3] = xaA[3]; }
    const float2 t0B[4] = {xaA[0], xaA[2], xbA[2], xbA[0]};   // thread 0: X[128], X[384], X[640], X[896]
    float2 rcA[4], rcB[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if constexpr ((AT_STFT_ABL & 24) == 0) {
      rcA[m] = to_partner(t0 ? xbB[3 - m] : xbA[3 - m]);      // bins 256 m + 192 + lane
      rcB[m] = to_partner(t0 ? t0B[m] : xbB[3 - m]);          // bins 256 m + 128 + lane
      } else if constexpr ((AT_STFT_ABL & 16) == 0) { rcA[m] = to_partner(xbA[3 - m]); rcB[m] = to_partner(xbB[3 - m]); }
      else if constexpr ((AT_STFT_ABL & 8) == 0) { rcA[m] = t0 ? xbB[3 - m] : xbA[3 - m]; rcB[m] = t0 ? t0B[m] : xbB[3 - m]; }
      else { rcA[m] = xbA[3 - m]; rcB[m] = xbB[3 - m]; }
    }
    float2* __restrict__ orow = A.out + ((int64_t)row * n_out + fo) * (M + 1);
    auto cabs_v2 = [](float2 z) __attribute__((always_inline)) -> float {
      if constexpr ((AT_STFT_ABL & 1) != 0) return fmaf(z.x, z.x, z.y * z.y); else return cabs_fast(z);
    };
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k0 = 256 * m + t;
      st2<POL>(orow + k0, ascA[m]);
      st2<POL>(orow + k0 + 64, xaB[m]);
      st2<POL>(orow + k0 + 128, rcB[m]);
      st2<POL>(orow + k0 + 192, rcA[m]);
      if constexpr (MEL) {
        if constexpr ((AT_STFT_ABL & 4) == 0) {
        magbuf[k0 + 4 * (k0 >> 4)] = cabs_v2(ascA[m]);
        magbuf[k0 + 64 + 4 * ((k0 + 64) >> 4)] = cabs_v2(xaB[m]);
        magbuf[k0 + 128 + 4 * ((k0 + 128) >> 4)] = cabs_v2(rcB[m]);
        magbuf[k0 + 192 + 4 * ((k0 + 192) >> 4)] = cabs_v2(rcA[m]);
        }
      }
    }
    if constexpr (STATIC_STORES) {
      const float nyq0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nyq.x), 0));
      st2<POL>(orow + M, make_float2(nyq0, 0.f));   // every lane, one address: counted on every path
      // last magnitude row: the Nyquist bin and 15 zero columns (see the generic kernel)
      if constexpr (MEL) { if (t < 16) magbuf[(M / 16) * MAG_ROW + t] = t0 ? fabsf(nyq0) : 0.f; }
    } else {
      if (t0) {
        st2<POL>(orow + M, nyq);
        if constexpr (MEL) magbuf[M + 4 * (M >> 4)] = fabsf(nyq.x);
      }
      if constexpr (MEL) { if (t >= 1 && t < 16) magbuf[(M / 16) * MAG_ROW + t] = 0.f; }
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (MEL && PIPE == 0) {
      wave_sync();
      float* mrow = A.mel + ((int64_t)row * n_out + fo) * A.n_mels;
      MelOps o;
      load_unit(o, 0, 0);
      load_unit(o, 1, 1);
      mel_rounds(o, mrow, std::true_type{});
      wave_sync();
      if constexpr (STATIC_STORES) {
        melout_store(mrow);
        wave_sync();
      }
    }
    return have_nxt;
    }   // !FLOOR
  };

#if AT_DEV_KNOBS
  // flags bits 10-13 = k + 1: only the workgroups of XCD k (blockIdx % 8, the observed round-robin placement) work, the
  // others leave -- with AT_STFT_NX=1 that XCD then writes every 8th group of runs of the whole range (tools/regime.py --affinity)
  if (((A.flags >> 10) & 15) != 0 && (int)(blockIdx.x % 8) != ((A.flags >> 10) & 15) - 1) return;
  // development schedules (tools/regime.py): flags bits 8-9 = 1: every XCD span starts its rounds at a different phase
  // (span k begins k/8 of the way through and wraps); 2: spans are row-interleaved (span k owns rows k, k + 8, ...)
  const int span_mode = (A.flags >> 8) & 3;
  const int64_t n_rounds = (g_hi - g_lo + Wtot * RUN - 1) / (Wtot * RUN);
#else
  constexpr int span_mode = 0;
#endif
  for (int64_t rnd = 0;; ++rnd) {
    int64_t gbase = w0 + rnd * Wtot * RUN;
#if AT_DEV_KNOBS
    if (span_mode == 1) {
      if (rnd >= n_rounds) break;
      gbase = w0 + ((rnd + n_rounds * xcd / n_x) % n_rounds) * Wtot * RUN;
      if (gbase >= g_hi) continue;
    } else
#endif
    if (gbase >= g_hi) break;
    const int64_t g_end = min(gbase + RUN, g_hi);
    int64_t g = gbase;
    while (g < g_end) {
      // ---- first frame of a stretch: all 16 loads (or the edge path), phase 0
      int64_t row = g / gpr;
      int fo = (int)(g - row * gpr);
#if AT_DEV_KNOBS
      if (span_mode == 2) row = (row - g_lo / gpr) * n_x + xcd;     // (needs rows % n_x == 0: spans hold whole rows)
#endif
      const float* __restrict__ xr = A.x + row * A.T;
      int64_t s0 = ((int64_t)fo + A.frame_lo) * A.hop - M;
      if (s0 >= 0 && s0 + N <= Ti) {   // wave-uniform
        const float2* __restrict__ p2 = reinterpret_cast<const float2*>(xr + s0);
#pragma unroll
        for (int q = 0; q < 16; ++q) raw[q] = ld2<POL>(p2 + t + L * q);
      } else {
#pragma unroll 1
        for (int q = 0; q < 16; ++q) {
          const int n = t + L * q;
          fbuf[n] = make_float2(fetch_padded(xr, s0 + 2 * n, A.T, A.T2, 0, A.pad_mode),
                                fetch_padded(xr, s0 + 2 * n + 1, A.T, A.T2, 0, A.pad_mode));
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 16; ++q) raw[q] = fbuf[t + L * q];
        wave_sync();
      }
      bool more = frame_body(std::integral_constant<int, 0>{}, std::false_type{}, xr, row, fo, s0, g + 1 < g_end);
      ++g;
      // steady state: "4 loads, then the stores of one frame" in flight on every path; the phase
      // (which 4 sample registers are the newest) is a compile-time constant of each copy
      while (more) {
#define AT_NEXT(P)                                                                                   \
  ++fo; s0 += A.hop;                                                                                 \
  more = frame_body(std::integral_constant<int, P>{}, std::true_type{}, xr, row, fo, s0, g + 1 < g_end);               \
  ++g;
        AT_NEXT(1) if (!more) break;
        AT_NEXT(2) if (!more) break;
        AT_NEXT(3) if (!more) break;
        AT_NEXT(0)
#undef AT_NEXT
      }
      if constexpr (PIPE != 0) mel_drain(row, fo);
    }
  }
}

constexpr size_t v2_lds_floats(int n_units) {
  return (size_t)4 * 2 * WAVE_LDS_SLOTS + 64 * 36 + 64 * 20 + 16 * 36 + (size_t)n_units * (MELW_ROW + 2);
}

// Run-length balancing shared by both kernels: every wave of an XCD span gets the same number of
// whole runs (see launch_one).
static int balanced_run(int64_t total_groups, int64_t blocks, int nw, int run_max = AT_STFT_RUN, int n_xcd = 8,
                        int min_runs = 1) {
  const int64_t n_x = blocks < n_xcd ? blocks : n_xcd;
  const int64_t waves_x = (blocks / n_x) * nw;
  const int64_t span = (total_groups + n_x - 1) / n_x;
  const int64_t per_wave = (span + waves_x - 1) / (waves_x > 0 ? waves_x : 1);
  int64_t runs = (per_wave + run_max - 1) / run_max;
  if (runs < min_runs && per_wave >= 16 * (int64_t)min_runs) runs = min_runs;
  int64_t run = (per_wave + runs - 1) / (runs > 0 ? runs : 1);
  if (run < 1) run = 1;
  if (run > run_max) run = run_max;
  return (int)run;
}

template <int NR, int POL = 0, int VAR = 0>
int launch_v2(const StftArgs& A, int n_cu, hipStream_t stream) {
  constexpr int NW = 4;
  auto kern = stft_mel_kernel_v2<NR, POL, VAR>;
  const size_t bytes = (v2_lds_floats(A.n_units) + (((VAR >> 2) & 3) ? NW * 144 : 0)) * 4;   // PIPE: + the waves' band-sum regions
  if (bytes > 160 * 1024) return AT_ERR_UNSUPPORTED;
  int e = at::allow_big_lds(reinterpret_cast<const void*>(kern));
  if (e != AT_OK) return e;
  int per_cu = (int)((160 * 1024) / bytes);
  if (per_cu > 2) per_cu = 2;   // 2 waves per SIMD (launch bounds)
  int64_t blocks = (A.total_groups + NW - 1) / NW;
  if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
  StftArgs B = A;
  // at least three runs per wave when they stay >= 16 frames long (64 items: 3 x 18 frames measured 5 %
  // faster than 1 x 54)
  B.run = balanced_run(A.total_groups, blocks, NW, A.run_max, A.n_xcd, 3);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), bytes, stream, B);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

template <int M, int NW, bool VEC2, int NR>
int launch_one(const StftArgs& A, size_t lds_bytes, int max_blocks, hipStream_t stream) {
  auto kern = stft_mel_kernel<M, NW, VEC2, NR>;
  {
    int e = at::allow_big_lds(reinterpret_cast<const void*>(kern));
    if (e != AT_OK) return e;
  }
  int64_t blocks = (A.total_groups + NW - 1) / NW;
  if (blocks > max_blocks) blocks = max_blocks;
  // Run length: every wave of an XCD span should get the same number of whole runs.  With a fixed
  // run of 16 a small batch (54 frame groups per wave at B=64) leaves some waves with 4 runs and
  // others with 3 -- the launch then lasts 64/54 of its balanced time.
  StftArgs B = A;
  // (n_fft 1024: runs of 64 groups measured 2 % faster than 16 in two rounds, the smaller sizes alike or slower: s23)
  B.run = balanced_run(A.total_groups, blocks, NW, M >= 512 ? 4 * AT_STFT_RUN : AT_STFT_RUN);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), lds_bytes, stream, B);
  AT_LAUNCH_CHECK();
  return AT_OK;
}

template <int M, int NW, bool VEC2, int NR>
int launch_nr(const StftArgs& A, int n_cu, hipStream_t stream) {
  size_t fl = lds_fixed_floats<M>(NW) + (size_t)A.n_units * (MELW_ROW + 2) + runstore_mag_floats<M, NR>(NW);
  const size_t bytes = fl * 4;
  if (bytes > 160 * 1024) return AT_ERR_UNSUPPORTED;
  // persistent grid: as many blocks as are co-resident (LDS-limited; the kernel needs ~200 VGPRs,
  // i.e. 2 waves/SIMD = 8 waves/CU), times #CUs
  int per_cu = (int)((160 * 1024) / bytes);
  constexpr int WPS = M <= 256 ? AT_STFT_WPS_SMALL : AT_STFT_WPS;
  const int by_waves = (4 * WPS) / NW > 0 ? (4 * WPS) / NW : 1;
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  return launch_one<M, NW, VEC2, NR>(A, bytes, n_cu * per_cu, stream);
}

template <int M, int NW, bool VEC2>
int launch_mw(const StftArgs& A, int n_cu, hipStream_t stream) {
  switch (A.n_units / 64) {
    case 0: return launch_nr<M, NW, VEC2, 0>(A, n_cu, stream);
    case 2: return launch_nr<M, NW, VEC2, 2>(A, n_cu, stream);
    case 4: return launch_nr<M, NW, VEC2, 4>(A, n_cu, stream);
    case 6: return launch_nr<M, NW, VEC2, 6>(A, n_cu, stream);
  }
  return AT_ERR_UNSUPPORTED;
}

template <int M>
int launch_m(const StftArgs& A, bool vec2, int n_cu, hipStream_t stream) {
  return vec2 ? launch_mw<M, AT_STFT_NW, true>(A, n_cu, stream) : launch_mw<M, AT_STFT_NW, false>(A, n_cu, stream);
}

}  // namespace

using at::device_cu_count;
using at::env_int_once;

// Measurement knobs of the v2 schedule (development builds only -- at::env_int_once folds to the default in the shipped
// library): read once; AT_STFT_TUNE=1 re-reads them on every call so that one process can sweep them (tools/stftsweep.py,
// tools/regime.py).  The kernel and its zero-compute twin take the same values.
struct V2Tuning { int flags, run_max, n_xcd, stagger; };
static V2Tuning v2_tuning() {
  auto read_tuning = [] {
    return V2Tuning{env_int_once("AT_STFT_FLAGS", -1), env_int_once("AT_STFT_RUNMAX", AT_STFT_RUN_V2), env_int_once("AT_STFT_NX", 0),
                    env_int_once("AT_STFT_STAGGERV2", 0)};
  };
  static const int tune_each_call = env_int_once("AT_STFT_TUNE", 0);
  static V2Tuning tuning = read_tuning();
  if (tune_each_call) tuning = read_tuning();
  return tuning;
}

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

// 1 when the fused wave-FFT kernels of this file cover n_fft (powers of two, 32 ... 2048): fused mel,
// register reuse, the adjoint kernels of istft.hip.
int at_stft_fused_supported(int n_fft) {
  return (n_fft >= 32 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0) ? 1 : 0;
}

// 1 when SOME native kernel covers n_fft: the fused ones, or the generic mixed-radix transform
// (even n_fft <= 16384 whose half factors into 2, 3, 5: 4096, 8192, 400, 1200, 1920 ...).
int at_stft_native_supported(int n_fft) {
  if (at_stft_fused_supported(n_fft)) return 1;
  int radix[16];
  return at::generic_fft_plan(n_fft, radix) > 0 ? 1 : 0;
}

// Host helper: compress a dense (n_mels, n_bins) float32 filterbank into the unit tables
// consumed by at_stft_mel_f32.  Call with unit_info == NULL to get the unit count (a multiple
// of 64); then with unit_info[2*n] ints and unit_w[16*n] floats.  Returns the unit count, or a
// negative error (AT_ERR_UNSUPPORTED if a band needs more than 8 rows of 16 bins).
int at_mel_units_host(const float* basis, int n_mels, int n_bins, int* unit_info, float* unit_w) {
  if (!basis || n_mels <= 0 || n_mels >= 0xffff || n_bins <= 0) return AT_ERR_INVALID;
  auto pad_unit = [&](int n) {
    // a padding unit reads the magnitude row of its predecessor (all-zero weights): the same LDS address as its neighbour
    // is a broadcast, while row 0 -- what padding pointed at until round 6 -- shares its banks with rows 16, 32, 48 (row
    // pitch 20 floats) and cost the unit reads of the 80-band table 20 conflict cycles per frame (84 instead of 64)
    const int row = n > 0 ? (unit_info[2 * (n - 1)] & 0xffff) : 0;
    unit_info[2 * n] = row | (0xffff << 16);
    unit_info[2 * n + 1] = 0;
    for (int i = 0; i < 16; ++i) unit_w[16 * n + i] = 0.f;
  };
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const bool fill = pass == 1;
    n = 0;
    for (int m = 0; m < n_mels; ++m) {
      const float* b = basis + (int64_t)m * n_bins;
      int lo = -1, hi = -1;
      for (int k = 0; k < n_bins; ++k)
        if (b[k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
      int r0 = 0, r1 = 0;  // all-zero band: one unit of zeros so that the output is written
      if (lo >= 0) { r0 = lo / 16; r1 = hi / 16; }
      const int cnt = r1 - r0 + 1;
      if (cnt > 16) return AT_ERR_UNSUPPORTED;
      while ((n % 16) + cnt > 16) {  // a band never straddles a 16-lane DPP row
        if (fill) pad_unit(n);
        ++n;
      }
      for (int j = 0; j < cnt; ++j, ++n) {
        if (!fill) continue;
        const int r = r0 + j;
        unit_info[2 * n] = r | (m << 16);
        // shift-left tree over `cnt` adjacent lanes: lane j adds lane j+d when j % (2d) == 0 and j+d < cnt
        int flags = 0;
        for (int step = 0, d = 1; step < 4; ++step, d <<= 1)
          if ((j % (2 * d)) == 0 && j + d < cnt) flags |= 1 << step;
        if (j == 0) flags |= 16;  // this lane stores the band
        unit_info[2 * n + 1] = flags;
        for (int i = 0; i < 16; ++i) {
          const int k = 16 * r + i;
          unit_w[16 * n + i] = (k < n_bins) ? b[k] : 0.f;
        }
      }
    }
    // the kernel is instantiated for 2, 4 or 6 rounds of 64 units
    const int target = n <= 128 ? 128 : (n <= 256 ? 256 : 384);
    if (n > 384) return AT_ERR_UNSUPPORTED;
    while (n < target) {
      if (fill) pad_unit(n);
      ++n;
    }
    if (!unit_info || !unit_w) return n;
  }
  return n;
}

// Host helper: the BANDED form of a dense (n_mels, n_bins) filterbank for the generic-size kernel: every row is cut to
// its non-zero span (start rounded down to a multiple of 4 bins) and the span into CHUNKS of 16 bins.  Call with info == NULL to get the chunk count n; then
// info[n + 2 * n_mels] = {first bin of every chunk} followed by {first chunk, chunk count} per band, and w[16 * n] = the
// chunk weights, zero padded (an all-zero row has no chunk).
int at_mel_bands_host(const float* basis, int n_mels, int n_bins, int* info, float* w) {
  if (!basis || n_mels <= 0 || n_bins <= 0) return AT_ERR_INVALID;
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const bool fill = pass == 1;
    const int total = n;
    n = 0;
    for (int m = 0; m < n_mels; ++m) {
      const float* b = basis + (int64_t)m * n_bins;
      int lo = -1, hi = -1;
      for (int k = 0; k < n_bins; ++k)
        if (b[k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
      if (lo >= 0) lo &= ~3;              // chunks start at multiples of 4 bins: 16-byte LDS reads of the magnitudes
      const int cnt = lo < 0 ? 0 : (hi - lo + 16) / 16;
      if (fill) { info[total + 2 * m] = n; info[total + 2 * m + 1] = cnt; }
      for (int c = 0; c < cnt; ++c, ++n) {
        if (!fill) continue;
        info[n] = lo + 16 * c;
        for (int j = 0; j < 16; ++j) {
          const int k = lo + 16 * c + j;
          w[16 * n + j] = (k <= hi) ? b[k] : 0.f;
        }
      }
    }
    if (!info || !w) return n;
  }
  return n;
}

int at_stft_mel_f32(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles,
                    int n_fft, int hop, int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out,
                    float* stft_out, const int* mel_unit_info, const float* mel_unit_w, int n_units, int n_mels,
                    float* mel_out, void* stream) {
  if (rows == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !window || !twiddles || rows < 0 || T <= 0 || hop <= 0 || pad < 0 || right_pad < 0 ||
      frame_lo < 0 || n_frames_out < 0)
    return AT_ERR_INVALID;
  if (!at_stft_native_supported(n_fft)) return AT_ERR_UNSUPPORTED;
  const bool mel = mel_out != nullptr;
  if (!stft_out) return AT_ERR_INVALID;  // stft_data is always produced (audio_signal.py:1210)
  if (!at_stft_fused_supported(n_fft)) {
    // generic sizes: mixed-radix workgroup FFT (csrc/stft_generic.hip).  Its fused mel stage takes the filterbank in
    // BANDED form (at_mel_bands_host): mel_unit_info = n_units first bins of 16-bin chunks, then {first chunk, count} per
    // band; mel_unit_w = (n_units, 16) zero-padded chunk weights.
    if (mel && (!mel_unit_info || !mel_unit_w || n_units <= 0 || n_mels <= 0 || n_fft / 2 > 4096)) return AT_ERR_UNSUPPORTED;
    const int64_t T2g = T + 2 * (int64_t)pad + right_pad;
    if (n_fft / 2 >= T2g) return AT_ERR_INVALID;
    if (pad_mode == at::PAD_REFLECT && (pad >= T || pad + right_pad >= T)) return AT_ERR_INVALID;
    if (frame_lo + n_frames_out > 1 + T2g / hop) return AT_ERR_INVALID;
    if (T >= (1LL << 31) || n_frames_out >= (1LL << 31) / (n_fft / 2 + 1)) return AT_ERR_UNSUPPORTED;
    if (n_frames_out == 0) return AT_OK;
    return at::stft_generic(x, rows, T, window, twiddles, n_fft, hop, pad, right_pad, pad_mode, frame_lo, n_frames_out,
                            stft_out, mel_unit_info, mel ? mel_unit_info + n_units : nullptr, mel_unit_w, n_units, n_mels,
                            mel ? mel_out : nullptr, reinterpret_cast<hipStream_t>(stream));
  }
  if (mel && (!mel_unit_info || !mel_unit_w || (n_units != 128 && n_units != 256 && n_units != 384) || n_mels <= 0))
    return AT_ERR_INVALID;
  const int M = n_fft / 2;
  const int64_t T2 = T + 2 * (int64_t)pad + right_pad;
  if (M >= T2) return AT_ERR_INVALID;  // torch.stft reflect padding needs n_fft/2 < length
  if (pad_mode == at::PAD_REFLECT && (pad >= T || pad + right_pad >= T)) return AT_ERR_INVALID;
  const int64_t n_total = 1 + T2 / hop;
  if (frame_lo + n_frames_out > n_total) return AT_ERR_INVALID;
  if (T >= (1LL << 31) || n_frames_out >= (1LL << 31) / (M + 1)) return AT_ERR_UNSUPPORTED;
  if (rows == 0 || n_frames_out == 0) return AT_OK;
  const int FW = 64 / (M / 16);
  // mel stage limits: <= 6 rounds of 64 units; FW frames of padded magnitudes in the wave slab
  if (mel && FW * (M / 16 + 1) * MAG_ROW > 2 * WAVE_LDS_SLOTS) return AT_ERR_UNSUPPORTED;

  StftArgs A;
  A.x = x; A.window = window; A.tw = reinterpret_cast<const float2*>(twiddles);
  A.out = reinterpret_cast<float2*>(stft_out); A.mel = mel_out;
  A.unit_info = mel_unit_info; A.unit_w = mel_unit_w;
  A.T = T; A.rows = rows; A.n_out = n_frames_out; A.frame_lo = frame_lo; A.hop = hop; A.pad = pad;
  A.T2 = T2; A.pad_mode = pad_mode; A.n_units = mel ? n_units : 0; A.n_mels = n_mels;
  A.groups_per_row = (int)((n_frames_out + FW - 1) / FW);
  A.total_groups = rows * A.groups_per_row;
  static const int dbg_mode = AT_STFT_DEBUGMODES ? env_int_once("AT_STFT_DEBUG", 0) : 0;
  static const int allow_reuse = env_int_once("AT_STFT_REUSE", 1);
  static const int use_v2 = env_int_once("AT_STFT_V2", 1);
  const V2Tuning tuning = v2_tuning();
  // default cache policy: `nt` stores of the spectrum when the mel stage runs (3-4 % faster on both
  // boxes measured), plain stores otherwise (nt: +6 % on one box, -2 % on the other)
  A.flags = tuning.flags >= 0 ? tuning.flags : (mel ? 1 : 0);
  A.run_max = tuning.run_max < 1 ? 1 : tuning.run_max;
  // XCD spans of the v2 schedule: one per XCD of the device.  Workgroups are dealt round-robin over the
  // XCDs (an observed placement, not an API guarantee; the spans only affect L2 locality at run
  // boundaries, never results).  An XCD has 32 CUs, so a partitioned device (CPX: 32 CUs) gets one span.
  {
    int auto_x = device_cu_count() / 32;
    auto_x = auto_x < 1 ? 1 : (auto_x > 8 ? 8 : auto_x);
    A.n_xcd = tuning.n_xcd >= 1 ? tuning.n_xcd : auto_x;
  }
  A.stagger = tuning.stagger < 0 ? 0 : tuning.stagger;
  A.debug = dbg_mode;
  A.reuse_shift = (FW == 1 && hop % (2 * (M / 16)) == 0 && pad == 0) ? hop / (2 * (M / 16)) : 0;
  if (!allow_reuse) A.reuse_shift = 0;
  const bool vec2 = ((T % 2) == 0 && (hop % 2) == 0 && (M % 2) == 0 && (reinterpret_cast<uintptr_t>(x) % 8) == 0);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int n_cu = device_cu_count();
  if (use_v2 && M == 1024 && vec2 && A.reuse_shift == 4 && A.debug == 0 && T >= 2 * 2048) {
    // the reference's default transform at 44.1 / 48 kHz: n_fft 2048, hop 512, no match_stride
    // flags bit 0: nt spectrum stores; bit 5: the round-2 store code (conditional Nyquist / band stores) instead
    // of the static-store-count variant.  The gathered band stores cover n_mels <= 128.
    // (session s40, same box, interleaved: with mel 2.172 vs 2.190 ms nt / 2.134 vs 2.163 ms plain stores; without
    // mel the round-2 code is 0.9 % faster, 1.814 vs 1.830 ms -- there the only uncounted store was the Nyquist one)
    const bool stat = !(A.flags & 32) && A.n_units != 0 && A.n_mels <= 128;
    constexpr int HXV = 16 * AT_STFT_V2_HX;       // half exchange (every v2 variant)
    constexpr int SV = 1 + 4 * AT_STFT_V2_PIPE + HXV;   // static store count (+ the pipelined mel stage)
    switch (A.n_units / 64) {
#define AT_V2_POL(NRV, SV)                                                                   \
  if (stat) return (A.flags & 1) ? launch_v2<NRV, 1, SV>(A, n_cu, s) : launch_v2<NRV, 0, SV>(A, n_cu, s); \
  return (A.flags & 1) ? launch_v2<NRV, 1, HXV>(A, n_cu, s) : launch_v2<NRV, 0, HXV>(A, n_cu, s);
      case 0: AT_V2_POL(0, 1 + HXV)
      case 2: return stat ? launch_v2<2, 0, SV>(A, n_cu, s) : launch_v2<2, 0, HXV>(A, n_cu, s);
      case 4: AT_V2_POL(4, SV)
#undef AT_V2_POL
      case 6: return stat ? launch_v2<6, 0, SV>(A, n_cu, s) : launch_v2<6, 0, HXV>(A, n_cu, s);
    }
  }
  switch (M) {
    case 16: return launch_m<16>(A, vec2, n_cu, s);
    case 32: return launch_m<32>(A, vec2, n_cu, s);
    case 64: return launch_m<64>(A, vec2, n_cu, s);
    case 128: return launch_m<128>(A, vec2, n_cu, s);
    case 256: return launch_m<256>(A, vec2, n_cu, s);
    case 512: return launch_m<512>(A, vec2, n_cu, s);
    case 1024: return launch_m<1024>(A, vec2, n_cu, s);
  }
  return AT_ERR_UNSUPPORTED;
}


#if AT_DEV_KNOBS   // (development library only since round 6: VERDICT r05 weak #12)
// Measurement twin of at_stft_mel_f32 for the n_fft 2048 / hop 512 kernel (same arguments): launches
// stft_mel_kernel_v2<..., FLOOR> -- identical grid, schedule, addresses, load / store instruction sequence and cache
// policy, no transform, no mel arithmetic -- so that its duration is the zero-compute cost of the kernel's traffic on
// THIS device (bench.py: roofline.floor_ms).  The outputs are overwritten with meaningless values.
// AT_ERR_UNSUPPORTED for every shape the v2 kernel does not take (then there is no floor to print).
int at_stft_mel_floor_f32(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles,
                          int n_fft, int hop, int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out,
                          float* stft_out, const int* mel_unit_info, const float* mel_unit_w, int n_units, int n_mels,
                          float* mel_out, void* stream) {
  if (rows == 0) return AT_OK;
  if (!x || !window || !twiddles || !stft_out || rows < 0 || T <= 0 || frame_lo != 0 || n_frames_out <= 0) return AT_ERR_INVALID;
  const bool mel = mel_out != nullptr;
  if (n_fft != 2048 || hop != 512 || pad != 0 || right_pad != 0 || T < 2 * 2048 || (T & 1) ||
      (reinterpret_cast<uintptr_t>(x) % 8) != 0 || n_frames_out != 1 + T / hop ||
      (mel && (n_units != 256 || n_mels <= 0 || n_mels > 128 || !mel_unit_info || !mel_unit_w)))
    return AT_ERR_UNSUPPORTED;
  if (T >= (1LL << 31) || n_frames_out >= (1LL << 31) / 1025) return AT_ERR_UNSUPPORTED;
  StftArgs A;
  A.x = x; A.window = window; A.tw = reinterpret_cast<const float2*>(twiddles);
  A.out = reinterpret_cast<float2*>(stft_out); A.mel = mel_out;
  A.unit_info = mel_unit_info; A.unit_w = mel_unit_w;
  A.T = T; A.rows = rows; A.n_out = n_frames_out; A.frame_lo = 0; A.hop = hop; A.pad = 0;
  A.T2 = T; A.pad_mode = pad_mode; A.n_units = mel ? n_units : 0; A.n_mels = n_mels;
  A.groups_per_row = (int)n_frames_out;
  A.total_groups = rows * A.groups_per_row;
  const V2Tuning tuning = v2_tuning();
  A.flags = mel ? 1 : 0;
  if (tuning.flags >= 0) A.flags = tuning.flags;
  A.run_max = tuning.run_max < 1 ? 1 : tuning.run_max;
  int auto_x = device_cu_count() / 32;
  auto_x = auto_x < 1 ? 1 : (auto_x > 8 ? 8 : auto_x);
  A.n_xcd = tuning.n_xcd >= 1 ? tuning.n_xcd : auto_x;
  A.stagger = 0; A.debug = 0; A.reuse_shift = 4; A.run = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (mel && tuning.flags >= 0 && !(tuning.flags & 1)) return launch_v2<4, 0, 3>(A, device_cu_count(), s);   // twin with plain spectrum stores
  return mel ? launch_v2<4, 1, 3>(A, device_cu_count(), s) : launch_v2<0, 0, 2>(A, device_cu_count(), s);
}
#endif   // AT_DEV_KNOBS

}  // extern "C"
