// Mixed-radix workgroup FFT used for the transform sizes the wave-FFT kernels do not cover (n_fft = 4096 ...
// 16384, even non-power-of-two windows such as 400 / 1200 / 1920) and by the four-step convolution.
// Prime radices 2, 3, 5, 7 plus the COMPOSITE radices 4, 8, 9, 16, 25 (round 3): a pass costs two workgroup
// barriers and one LDS round trip of every point whatever its radix, so 2000 = 25 * 5 * 16 in three passes
// instead of 5 * 5 * 5 * 4 * 4 in five removes 40 % of the barriers and of the LDS traffic of a row transform.
#pragma once
#include "at_common.h"

namespace at {

// radix butterflies shared by the workgroup FFT kernels (stft_generic.hip, longconv.hip)
namespace gfft {
constexpr int MAX_RADIX = 25;

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// forward DFT of R points in place
template <int R>
__device__ __forceinline__ void dft_r(float2 (&v)[MAX_RADIX]);
template <>
__device__ __forceinline__ void dft_r<2>(float2 (&v)[MAX_RADIX]) {
  const float2 a = v[0], b = v[1];
  v[0] = make_float2(a.x + b.x, a.y + b.y);
  v[1] = make_float2(a.x - b.x, a.y - b.y);
}
template <>
__device__ __forceinline__ void dft_r<4>(float2 (&v)[MAX_RADIX]) {
  const float2 t0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
  const float2 t2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), t3 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
  v[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
  v[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
  v[1] = make_float2(t1.x + t3.y, t1.y - t3.x);   // t1 - i t3
  v[3] = make_float2(t1.x - t3.y, t1.y + t3.x);   // t1 + i t3
}
template <>
__device__ __forceinline__ void dft_r<3>(float2 (&v)[MAX_RADIX]) {
  const float S3 = 0.86602540378443864676f;       // sin(2 pi / 3)
  const float2 s = make_float2(v[1].x + v[2].x, v[1].y + v[2].y);
  const float2 d = make_float2(v[1].x - v[2].x, v[1].y - v[2].y);
  const float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
  v[0] = make_float2(v[0].x + s.x, v[0].y + s.y);
  // X1 = m - i S3 d,  X2 = m + i S3 d
  v[1] = make_float2(m.x + S3 * d.y, m.y - S3 * d.x);
  v[2] = make_float2(m.x - S3 * d.y, m.y + S3 * d.x);
}
template <>
__device__ __forceinline__ void dft_r<5>(float2 (&v)[MAX_RADIX]) {
  const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;   // cos(2 pi/5), cos(4 pi/5)
  const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;    // sin(2 pi/5), sin(4 pi/5)
  const float2 a1 = make_float2(v[1].x + v[4].x, v[1].y + v[4].y), b1 = make_float2(v[1].x - v[4].x, v[1].y - v[4].y);
  const float2 a2 = make_float2(v[2].x + v[3].x, v[2].y + v[3].y), b2 = make_float2(v[2].x - v[3].x, v[2].y - v[3].y);
  const float2 x0 = v[0];
  v[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
  const float2 p1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x, x0.y + C1 * a1.y + C2 * a2.y);
  const float2 p2 = make_float2(x0.x + C2 * a1.x + C1 * a2.x, x0.y + C2 * a1.y + C1 * a2.y);
  const float2 q1 = make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
  const float2 q2 = make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
  // X_k = p - i q  (k = 1, 2),  X_{5-k} = p + i q
  v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);
  v[4] = make_float2(p1.x - q1.y, p1.y + q1.x);
  v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);
  v[3] = make_float2(p2.x - q2.y, p2.y + q2.x);
}

template <>
__device__ __forceinline__ void dft_r<7>(float2 (&v)[MAX_RADIX]) {
  const float C1 = 0.62348980185873353053f, C2 = -0.22252093395631440429f, C3 = -0.90096886790241912624f;  // cos(2 pi m/7)
  const float S1 = 0.78183148246802980871f, S2 = 0.97492791218182360702f, S3 = 0.43388373911755812048f;   // sin(2 pi m/7)
  const float2 a1 = make_float2(v[1].x + v[6].x, v[1].y + v[6].y), b1 = make_float2(v[1].x - v[6].x, v[1].y - v[6].y);
  const float2 a2 = make_float2(v[2].x + v[5].x, v[2].y + v[5].y), b2 = make_float2(v[2].x - v[5].x, v[2].y - v[5].y);
  const float2 a3 = make_float2(v[3].x + v[4].x, v[3].y + v[4].y), b3 = make_float2(v[3].x - v[4].x, v[3].y - v[4].y);
  const float2 x0 = v[0];
  v[0] = make_float2(x0.x + a1.x + a2.x + a3.x, x0.y + a1.y + a2.y + a3.y);
  const float2 p1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x + C3 * a3.x, x0.y + C1 * a1.y + C2 * a2.y + C3 * a3.y);
  const float2 p2 = make_float2(x0.x + C2 * a1.x + C3 * a2.x + C1 * a3.x, x0.y + C2 * a1.y + C3 * a2.y + C1 * a3.y);
  const float2 p3 = make_float2(x0.x + C3 * a1.x + C1 * a2.x + C2 * a3.x, x0.y + C3 * a1.y + C1 * a2.y + C2 * a3.y);
  const float2 q1 = make_float2(S1 * b1.x + S2 * b2.x + S3 * b3.x, S1 * b1.y + S2 * b2.y + S3 * b3.y);
  const float2 q2 = make_float2(S2 * b1.x - S3 * b2.x - S1 * b3.x, S2 * b1.y - S3 * b2.y - S1 * b3.y);
  const float2 q3 = make_float2(S3 * b1.x - S1 * b2.x + S2 * b3.x, S3 * b1.y - S1 * b2.y + S2 * b3.y);
  // X_k = p_k - i q_k,  X_{7-k} = p_k + i q_k
  v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);
  v[6] = make_float2(p1.x - q1.y, p1.y + q1.x);
  v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);
  v[5] = make_float2(p2.x - q2.y, p2.y + q2.x);
  v[3] = make_float2(p3.x + q3.y, p3.y - q3.x);
  v[4] = make_float2(p3.x - q3.y, p3.y + q3.x);
}

// ---- composite radices: two in-register stages with constant twiddles (decimation in time:
//      n = R2 n1 + n2, k = k1 + R1 k2 with R = R1 R2)
__device__ __forceinline__ float2 cmul_c(float2 a, float c, float s) {   // a * (c - i s)
  return make_float2(fmaf(a.x, c, a.y * s), fmaf(a.y, c, -a.x * s));
}
template <>
__device__ __forceinline__ void dft_r<8>(float2 (&v)[MAX_RADIX]) {
  const float H = 0.70710678118654752440f;
  float2 e[MAX_RADIX], o[MAX_RADIX];
  e[0] = v[0]; e[1] = v[2]; e[2] = v[4]; e[3] = v[6];
  o[0] = v[1]; o[1] = v[3]; o[2] = v[5]; o[3] = v[7];
  dft_r<4>(e);
  dft_r<4>(o);
  o[1] = cmul_c(o[1], H, H);                              // w8^1
  o[2] = make_float2(o[2].y, -o[2].x);                    // w8^2 = -i
  o[3] = cmul_c(o[3], -H, H);                             // w8^3
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = make_float2(e[k].x + o[k].x, e[k].y + o[k].y);
    v[k + 4] = make_float2(e[k].x - o[k].x, e[k].y - o[k].y);
  }
}
template <>
__device__ __forceinline__ void dft_r<16>(float2 (&v)[MAX_RADIX]) {
  // 4 x 4: columns n2 = 0..3 hold n = 4 n1 + n2
  const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
  float2 col[4][MAX_RADIX];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) col[n2][n1] = v[4 * n1 + n2];
    dft_r<4>(col[n2]);
  }
  // twiddle w16^(n2 k1)
  col[1][1] = cmul_c(col[1][1], C1, S1); col[1][2] = cmul_c(col[1][2], H, H);  col[1][3] = cmul_c(col[1][3], S1, C1);
  col[2][1] = cmul_c(col[2][1], H, H);   col[2][2] = make_float2(col[2][2].y, -col[2][2].x); col[2][3] = cmul_c(col[2][3], -H, H);
  col[3][1] = cmul_c(col[3][1], S1, C1); col[3][2] = cmul_c(col[3][2], -H, H); col[3][3] = cmul_c(col[3][3], -C1, -S1);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float2 r[MAX_RADIX];
    r[0] = col[0][k1]; r[1] = col[1][k1]; r[2] = col[2][k1]; r[3] = col[3][k1];
    dft_r<4>(r);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = r[k2];
  }
}
template <>
__device__ __forceinline__ void dft_r<9>(float2 (&v)[MAX_RADIX]) {
  // cos / sin of 2 pi m / 9, m = 1, 2, 4
  const float C1 = 0.76604444311897803520f, S1 = 0.64278760968653932632f;
  const float C2 = 0.17364817766693034885f, S2 = 0.98480775301220805937f;
  const float C4 = -0.93969262078590838405f, S4 = 0.34202014332566873304f;
  float2 col[3][MAX_RADIX];
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) {
#pragma unroll
    for (int n1 = 0; n1 < 3; ++n1) col[n2][n1] = v[3 * n1 + n2];
    dft_r<3>(col[n2]);
  }
  col[1][1] = cmul_c(col[1][1], C1, S1); col[1][2] = cmul_c(col[1][2], C2, S2);
  col[2][1] = cmul_c(col[2][1], C2, S2); col[2][2] = cmul_c(col[2][2], C4, S4);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    float2 r[MAX_RADIX];
    r[0] = col[0][k1]; r[1] = col[1][k1]; r[2] = col[2][k1];
    dft_r<3>(r);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) v[k1 + 3 * k2] = r[k2];
  }
}
template <>
__device__ __forceinline__ void dft_r<25>(float2 (&v)[MAX_RADIX]) {
  // w25^m = (cos, -sin)(2 pi m / 25) for the exponents n2 k1 in {1,2,3,4,6,8,9,12,16}
  constexpr float C[17] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899666f,
                           0.30901699437494742f, 0.06279051952931337f, -0.18738131458572463f, -0.42577929156507272f,
                           -0.63742398974868975f, -0.80901699437494742f, -0.92977648588825146f, -0.99211470131447788f,
                           -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494742f, -0.63742398974868975f};
  constexpr float S[17] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868873f, 0.84432792550201508f,
                           0.95105651629515357f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601958f,
                           0.77051324277578925f, 0.58778525229247313f, 0.36812455268467797f, 0.12533323356430426f,
                           -0.12533323356430426f, -0.36812455268467797f, -0.58778525229247313f, -0.77051324277578925f};
  float2 col[5][MAX_RADIX];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
#pragma unroll
    for (int n1 = 0; n1 < 5; ++n1) col[n2][n1] = v[5 * n1 + n2];
    dft_r<5>(col[n2]);
  }
#pragma unroll
  for (int n2 = 1; n2 < 5; ++n2)
#pragma unroll
    for (int k1 = 1; k1 < 5; ++k1) col[n2][k1] = cmul_c(col[n2][k1], C[n2 * k1], S[n2 * k1]);
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    float2 r[MAX_RADIX];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) r[n2] = col[n2][k1];
    dft_r<5>(r);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[k1 + 5 * k2] = r[k2];
  }
}

// ---- in-place mixed-radix passes over an LDS tile, shared by the four-step convolution (longconv.hip) and the
//      generic-size STFT (stft_generic.hip)
constexpr int TILE_POINTS = 4096;   // complex points of one LDS tile (colfft) / one row pair (rowconv)
constexpr int MAX_PASSES = 12;

struct PassList {
  int n;
  int radix[MAX_PASSES];
  int ns[MAX_PASSES];       // product of the radices before this pass
};


inline bool factor(int n, PassList* p) {
  p->n = 0;
  int ns = 1;
  auto take = [&](int r) {
    while (n % r == 0) {
      if (p->n == MAX_PASSES) return false;
      p->radix[p->n] = r; p->ns[p->n] = ns; ++p->n;
      ns *= r; n /= r;
    }
    return true;
  };
  // Odd radices first: a Stockham pass writes runs of NS consecutive points at a stride of NS R, and
  // with NS = 1, 4, 16 and R = 4 that is an 8-way LDS bank conflict in the row layout (36 % of the
  // LDS cycles of rowconv_kernel); strides of 3, 5, 7 points are conflict-free, and once the odd part
  // is done NS is large enough for the power-of-two passes.  Composite radices (25, 9, 16, 8) before
  // their primes: 2000 = 25 * 5 * 16 is three passes (six barriers) instead of five (ten).
  return take(25) && take(5) && take(9) && take(3) && take(7) && take(16) && take(8) && take(4) && take(2) && n == 1;
}

// ---------------------------------------------------------------- in-place mixed-radix passes
struct ColLayout {          // tile[point][column]
  int lcw, cmask;
  __device__ __forceinline__ void split(int id, int nb, int& batch, int& j) const { batch = id & cmask; j = id >> lcw; (void)nb; }
  __device__ __forceinline__ int addr(int batch, int p) const { return (p << lcw) + batch; }
};
struct RowLayout {          // buf[row slot][point]
  int N;
  // sw != 0 (N a multiple of 16): XOR swizzle of the low 4 bits with the next 4 -- a permutation inside every
  // 16-block, so contiguous reads stay conflict-free, and the stride-16 stores of a first radix-16 pass (NS = 1:
  // lane j writes points 16 j + q) land on 16 distinct bank pairs instead of one (65 % of the LDS cycles of the
  // first tiled STFT kernel were bank conflicts, profiles/r03_notes.md)
  int sw;
  __device__ __forceinline__ void split(int id, int nb, int& batch, int& j) const { batch = id >= nb ? 1 : 0; j = id - (batch ? nb : 0); }
  __device__ __forceinline__ int addr(int batch, int p) const { return batch * N + (sw ? (p ^ ((p >> 4) & 15)) : p); }
};
struct RowLayoutN {         // buf[row slot][point], any number of rows (the run-time-plan STFT tile: up to 64 short frames)
  int N, sw;
  // id / nb through a float reciprocal: exact for id < 2^13 (see stft_generic_tiled_kernel)
  __device__ __forceinline__ void split(int id, int nb, int& batch, int& j) const {
    batch = (int)(((float)id + 0.5f) * (1.0f / (float)nb));
    j = id - batch * nb;
  }
  __device__ __forceinline__ int addr(int batch, int p) const { return batch * N + (sw ? (p ^ ((p >> 4) & 15)) : p); }
};

// Per-pass twiddle blocks.  A pass of radix R behind NS = (product of the earlier radices) multiplies input q of the
// butterfly with k = j mod NS by w_{NS R}^{k q}.  Gathering these from ONE table w_N^t (index k q N / (NS R)) is a
// power-of-two stride across the lanes for every power-of-two NS R: up to 16-way LDS bank conflicts per read (half of
// rowconv's LDS cycles).  The blocks hold, for every pass with NS > 1, the (R - 1) x NS values in the order the lanes
// read them -- [q - 1][k], consecutive k in consecutive lanes -- one after the other: sum NS (R - 1) < N entries, the
// size of the plain table.
template <int NT>
__device__ __forceinline__ void build_pass_twiddles(float2* __restrict__ dst, const float2* __restrict__ src, int src_stride,
                                                    int N, const PassList& pl) {
  int off = 0;
  for (int p = 0; p < pl.n; ++p) {
    const int R = pl.radix[p], NS = pl.ns[p];
    if (NS <= 1) continue;
    const int tstep = N / (NS * R), cnt = NS * (R - 1);
    for (int idx = threadIdx.x; idx < cnt; idx += NT) {
      const int q1 = idx / NS, k = idx - q1 * NS;
      dst[off + idx] = src[(int64_t)(k * (q1 + 1) * tstep) * src_stride];
    }
    off += cnt;
  }
}

// One Stockham pass of radix R over `total` butterflies (all batches), in place: every thread reads
// its NB butterflies, the workgroup meets, every thread writes.  twp = this pass's twiddle block
// (build_pass_twiddles): twp[(q - 1) NS + k] = w_{NS R}^{k q}; unused when NS == 1.
// The loop is branch-free: a thread index past the last butterfly is clamped to it, so the surplus
// lanes repeat that butterfly and store the same values to the same slots.  With predicated
// iterations every butterfly was its own basic block and its LDS reads were not issued before the
// previous butterfly had finished (the kernel ran at 46 % VALU and 26 % LDS utilisation).
// WAVE = true (NT = 64): the tile belongs to ONE wave (the wave-per-row-pair form of the long convolution): the thread
// index is the lane and "the workgroup meets" is a wave-level fence -- no s_barrier, the waves of a workgroup never wait
// for each other.
template <int R, int NB, int NT, class L, bool WAVE = false>
__device__ __forceinline__ void pass_inplace(float2* __restrict__ buf, const float2* __restrict__ twp, int N, int NS,
                                             int total, const L lay) {
  static_assert(!WAVE || NT == 64, "a wave-level pass has 64 threads");
  const int nb = N / R;
  const float inv_ns = 1.0f / (float)NS;
  float2 v[NB][MAX_RADIX];
  int o[NB], ob[NB];
  // opaque copy of the thread index: keeps the per-butterfly index arithmetic of every pass variant
  // from being hoisted out of the pass loop (that cost > 128 live registers and spills)
  int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int last = total - 1;
  // the last iteration is skipped by the waves that lie entirely past the end (wave-uniform branch)
  const bool tail = __builtin_amdgcn_readfirstlane(tid & ~63) + NT * (NB - 1) < total;
  auto load_one = [&](int b) __attribute__((always_inline)) {
    const int id = min(tid + NT * b, last);
    int batch, j;
    lay.split(id, nb, batch, j);
    const int jd = (int)(((float)j + 0.5f) * inv_ns);     // j / NS (exact: j < 2048)
    const int k = j - jd * NS;
#pragma unroll
    for (int q = 0; q < R; ++q) v[b][q] = buf[lay.addr(batch, j + nb * q)];
    if (NS > 1) {
#pragma unroll
      for (int q = 1; q < R; ++q) v[b][q] = cmulf(v[b][q], twp[(q - 1) * NS + k]);
    }
    o[b] = jd * NS * R + k;
    ob[b] = batch;
  };
#pragma unroll
  for (int b = 0; b < NB - 1; ++b) load_one(b);
  if (tail) load_one(NB - 1);
#pragma unroll
  for (int b = 0; b < NB - 1; ++b) dft_r<R>(v[b]);
  if (tail) dft_r<R>(v[NB - 1]);
  if constexpr (WAVE) wave_sync(); else __syncthreads();
#pragma unroll
  for (int b = 0; b < NB - 1; ++b) {
#pragma unroll
    for (int q = 0; q < R; ++q) buf[lay.addr(ob[b], o[b] + NS * q)] = v[b][q];
  }
  if (tail) {
#pragma unroll
    for (int q = 0; q < R; ++q) buf[lay.addr(ob[NB - 1], o[NB - 1] + NS * q)] = v[NB - 1][q];
  }
  if constexpr (WAVE) wave_sync(); else __syncthreads();
}

// the variant for the iteration count of this launch (uniform): NB = ceil(total / NT)
template <int R, int NT, class L, bool WAVE = false>
__device__ __forceinline__ void pass_dispatch(float2* buf, const float2* tw, int N, int NS, int total, const L lay) {
  // (a wave-level tile holds at most WAVE_TILE_POINTS points: the variants past 8 butterflies per lane are never needed,
  //  the host checks wave_pass_list_ok before it picks the wave form)
  constexpr int MAXB = ((WAVE ? 2048 : TILE_POINTS) / R + NT - 1) / NT;
  const int nbi = (total + NT - 1) / NT;
  if (nbi <= 1) pass_inplace<R, 1, NT, L, WAVE>(buf, tw, N, NS, total, lay);
  else if (MAXB >= 2 && nbi == 2) pass_inplace<R, (MAXB >= 2 ? 2 : 1), NT, L, WAVE>(buf, tw, N, NS, total, lay);
  else if (MAXB >= 3 && nbi == 3) pass_inplace<R, (MAXB >= 3 ? 3 : 1), NT, L, WAVE>(buf, tw, N, NS, total, lay);
  else if (MAXB >= 4 && nbi == 4) pass_inplace<R, (MAXB >= 4 ? 4 : 1), NT, L, WAVE>(buf, tw, N, NS, total, lay);
  else if (MAXB >= 6 && nbi <= 6) pass_inplace<R, (MAXB >= 6 ? 6 : 1), NT, L, WAVE>(buf, tw, N, NS, total, lay);
  else if (MAXB >= 8) pass_inplace<R, (MAXB >= 8 ? 8 : 1), NT, L, WAVE>(buf, tw, N, NS, total, lay);
}

// the wave form runs at most 8 butterflies per lane and pass (pass_dispatch): batches * N / R <= 512 for every pass
inline bool wave_pass_list_ok(const PassList& pl, int N, int batches) {
  for (int p = 0; p < pl.n; ++p)
    if ((N / pl.radix[p]) * batches > 8 * 64) return false;
  return true;
}

template <int NT, class L, bool WAVE = false>
__device__ __forceinline__ void run_passes(float2* buf, const float2* twb /* pass blocks */, int N, const PassList& pl,
                                           int batches, const L lay) {
  int off = 0;
  for (int p = 0; p < pl.n; ++p) {
    const int R = pl.radix[p], NS = pl.ns[p];
    const int total = (N / R) * batches;
    const float2* tw = twb + off;
    switch (R) {
      case 16: pass_dispatch<16, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 25: pass_dispatch<25, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 8: pass_dispatch<8, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 9: pass_dispatch<9, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 4: pass_dispatch<4, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 2: pass_dispatch<2, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 3: pass_dispatch<3, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      case 5: pass_dispatch<5, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
      default: pass_dispatch<7, NT, L, WAVE>(buf, tw, N, NS, total, lay); break;
    }
    if (NS > 1) off += NS * (R - 1);
  }
}

}  // namespace gfft

// M = n_fft / 2 factors into {2, 3, 5, 7}: fills radix[] (composite radices 16, 8, 25, 9 first) and returns the pass count, 0 if not.
int generic_fft_plan(int n_fft, int* radix /* [16] */);

// STFT of `n_frames_out` frames per row (same argument meaning as at_stft_mel_f32, no mel).
// mel_out != null: fused banded mel epilogue (chunk tables of at_mel_bands_host: mel_chunk (n_chunks) first bins,
// mel_band (n_mels, 2) {first chunk, count}, mel_w (n_chunks, 16) zero-padded weights).
int stft_generic(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles, int n_fft, int hop,
                 int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out, float* stft_out,
                 const int* mel_chunk, const int* mel_band, const float* mel_w, int n_chunks, int n_mels, float* mel_out,
                 hipStream_t st);

// Fused inverse transform for n_fft 4096 / 8192 with hop = n_fft / 4 (stft_generic.hip: the hand-addressed tile run
// backwards, overlap-add in an LDS ring).  Argument meaning as at_istft_f32; `workspace`: istft_tiled_workspace_floats floats.
bool istft_tiled_supported(int n_fft, int hop);
int64_t istft_tiled_workspace_floats(int64_t n_frames, int n_fft, int hop);
int istft_tiled(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft, int hop,
                int lead, int64_t n_frames, int64_t length, float* out, float* workspace, hipStream_t st);

// One-pass inverse of the run-time sizes (speech windows ...): transform tiles + overlap-add in LDS (stft_generic.hip,
// istft_generic_ola_kernel).  lead = 0, every frame stored.  `workspace`: istft_generic_ola_workspace_floats floats (envelope).
bool istft_generic_ola_supported(int n_fft, int hop);
int64_t istft_generic_ola_workspace_floats(int64_t n_frames, int n_fft, int hop);
int istft_generic_ola(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles, int n_fft,
                      int hop, int64_t length, float* out, float* workspace, hipStream_t st);

// Inverse transform of every frame: X (rows, n_frames, n_fft/2+1) -> windowed frames (rows, n_frames, n_fft),
// the input of istft_ola_kernel.
int istft_frames_generic(const float* X, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                         int n_fft, float* frames, hipStream_t st);

}  // namespace at
