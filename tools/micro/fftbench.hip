// FFT-core microbenchmark for the "leaner wave FFT" question of DESIGN.md section 9: throughput of a
// 1024-point complex FFT resident in LDS/registers (no HBM traffic), two decompositions:
//   A  one wave per frame, 16 points per thread, radix 16.16.4 (the production fft_wave.h code),
//      wave-level exchanges, as resident as 255 / 168 / 128 VGPR would allow (2 / 3 / 4 waves per SIMD)
//   B  two waves per frame (workgroup of 128 threads), 8 points per thread, radix 8.8.8.2,
//      s_barrier exchanges, register-light
// Prints frames per microsecond for the whole GPU.   build: hipcc --offload-arch=gfx950 -O3 -o fftbench fftbench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../../audiotools_amd/csrc/fft_wave.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

// ---------------------------------------------------------------- A: production wave FFT
template <int WPS>
__global__ __launch_bounds__(256, WPS) void fft_a(const float2* __restrict__ tw, float2* __restrict__ out, int iters) {
  constexpr int M = 1024, L = 64, N = 2 * M;
  __shared__ float2 lds[4 * WAVE_LDS_SLOTS];
  __shared__ __attribute__((aligned(16))) float s_tw2[16 * 36];
  for (int i = threadIdx.x; i < 256; i += 256) {
    const int jj = i / 16, r = i % 16;
    reinterpret_cast<float2*>(s_tw2 + jj * 36)[r] = tw[r * jj * (N / 256)];
  }
  __syncthreads();
  const int t = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float2* fbuf = lds + wave * WAVE_LDS_SLOTS;
  float2 tw3b[4];
  for (int b = 0; b < 4; ++b) tw3b[b] = tw[((t + b * L) % 256) * (N / 1024)];
  float2 a[16];
  for (int q = 0; q < 16; ++q) a[q] = make_float2(0.001f * (t + 64 * q), 0.002f * q);
  for (int it = 0; it < iters; ++it) {
    pass_compute_store<16, 1, L>(a, fbuf, t, nullptr);
    wave_sync();
    load_points<L>(a, fbuf, t);
    wave_sync();
    {
      float2 tw2[16];
      const float2* rowp = reinterpret_cast<const float2*>(s_tw2 + (t & 15) * 36);
      for (int r = 1; r < 16; ++r) tw2[r] = rowp[r];
      pass_compute_store<16, 16, L>(a, fbuf, t, tw2);
    }
    wave_sync();
    load_points<L>(a, fbuf, t);
    wave_sync();
    {
      float2 tw3[16];
      for (int b = 0; b < 4; ++b) {
        tw3[b * 4 + 1] = tw3b[b];
        tw3[b * 4 + 2] = cmul(tw3b[b], tw3b[b]);
        tw3[b * 4 + 3] = cmul(tw3[b * 4 + 2], tw3b[b]);
      }
      pass_compute_store<4, 256, L>(a, fbuf, t, tw3);
    }
    wave_sync();
    load_points<L>(a, fbuf, t);   // result back into registers (the real kernels read it for the split step)
    wave_sync();
    for (int q = 0; q < 16; ++q) a[q] = make_float2(a[q].x * 0.03125f, a[q].y * 0.03125f);  // keep magnitudes bounded
  }
  float2 s = make_float2(0.f, 0.f);
  for (int q = 0; q < 16; ++q) s = cadd(s, a[q]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- B: 8 points per thread, 2 waves per frame
__device__ __forceinline__ int physb(int i) { return i ^ ((i >> 3) & 15); }

template <int R, int NS>
__device__ __forceinline__ void pass_b(float2 (&a)[8], float2* __restrict__ buf, int t, const float2* __restrict__ twp) {
  // 128 threads, 1024 points: 1024 / R butterflies -> NB per thread
  constexpr int NB = 8 / R;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = a[b + r * NB];
    if constexpr (NS > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) v[r] = cmul(v[r], twp[b * R + r]);
    }
    Dft<R>::run(v);
    const int j = t + b * 128;
    const int o0 = (j / NS) * (NS * R) + (j % NS);
#pragma unroll
    for (int r = 0; r < R; ++r) buf[physb(o0 + r * NS)] = v[r];
  }
}

__global__ __launch_bounds__(128) void fft_b(const float2* __restrict__ tw1024, float2* __restrict__ out, int iters) {
  __shared__ float2 buf[1024];
  const int t = threadIdx.x;
  // per-thread twiddles (loop invariant): pass 2 (NS = 8), pass 3 (NS = 64), pass 4 (R = 2, NS = 512, 4 butterflies)
  float2 tw2[8], tw3[8], tw4[8];
  for (int r = 1; r < 8; ++r) tw2[r] = tw1024[(r * (t % 8) * (1024 / 64)) & 1023];
  for (int r = 1; r < 8; ++r) tw3[r] = tw1024[(r * (t % 64) * (1024 / 512)) & 1023];
  for (int b = 0; b < 4; ++b) tw4[b * 2 + 1] = tw1024[((t + b * 128) % 512) & 1023];
  float2 a[8];
  for (int q = 0; q < 8; ++q) a[q] = make_float2(0.001f * (t + 128 * q), 0.002f * q);
  for (int it = 0; it < iters; ++it) {
    pass_b<8, 1>(a, buf, t, nullptr);
    __syncthreads();
    for (int q = 0; q < 8; ++q) a[q] = buf[physb(t + 128 * q)];
    __syncthreads();
    pass_b<8, 8>(a, buf, t, tw2);
    __syncthreads();
    for (int q = 0; q < 8; ++q) a[q] = buf[physb(t + 128 * q)];
    __syncthreads();
    pass_b<8, 64>(a, buf, t, tw3);
    __syncthreads();
    for (int q = 0; q < 8; ++q) a[q] = buf[physb(t + 128 * q)];
    __syncthreads();
    pass_b<2, 512>(a, buf, t, tw4);
    __syncthreads();
    for (int q = 0; q < 8; ++q) a[q] = buf[physb(t + 128 * q)];
    __syncthreads();
    for (int q = 0; q < 8; ++q) a[q] = make_float2(a[q].x * 0.03125f, a[q].y * 0.03125f);
  }
  float2 s = make_float2(0.f, 0.f);
  for (int q = 0; q < 8; ++q) s = cadd(s, a[q]);
  out[blockIdx.x * 128 + threadIdx.x] = s;
}

int main() {
  const int N = 2048;
  float2* h = new float2[N];
  for (int k = 0; k < N; ++k) h[k] = make_float2((float)cos(2 * M_PI * k / N), (float)-sin(2 * M_PI * k / N));
  float2* h1 = new float2[1024];
  for (int k = 0; k < 1024; ++k) h1[k] = make_float2((float)cos(2 * M_PI * k / 1024), (float)-sin(2 * M_PI * k / 1024));
  float2 *tw, *tw1, *out;
  CK(hipMalloc(&tw, N * 8)); CK(hipMemcpy(tw, h, N * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&tw1, 1024 * 8)); CK(hipMemcpy(tw1, h1, 1024 * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 64 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  float ms;
  auto report = [&](const char* name, int frames_per_block, int blocks) {
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double frames = (double)frames_per_block * blocks * iters;
    printf("%-44s %8.3f ms  %8.1f frames/us  (882688 frames = %.3f ms)\n", name, ms, frames / ms / 1e3, 882688.0 / (frames / ms));
    return 0;
  };
  for (int rep = 0; rep < 2; ++rep) {
    // A: blocks of 4 waves; WPS blocks-per-CU-worth resident (256 CUs)
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(fft_a<2>, dim3(256 * 2), dim3(256), 0, 0, tw, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    report("A 16 pts/thread, 2 waves/SIMD (production)", 4, 512);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(fft_a<3>, dim3(256 * 3), dim3(256), 0, 0, tw, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    report("A 16 pts/thread, 3 waves/SIMD (168 VGPR cap)", 4, 768);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(fft_a<4>, dim3(256 * 4), dim3(256), 0, 0, tw, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    report("A 16 pts/thread, 4 waves/SIMD (128 VGPR cap)", 4, 1024);
    for (int per_cu : {4, 8, 12, 16}) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(fft_b, dim3(256 * per_cu), dim3(128), 0, 0, tw1, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      char nm[64]; snprintf(nm, 64, "B 8 pts/thread, %d frames (%d waves) per CU", per_cu, 2 * per_cu);
      report(nm, 1, 256 * per_cu);
    }
  }
  return 0;
}
