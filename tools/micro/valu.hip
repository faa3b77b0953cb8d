// VALU issue-rate microbenchmark: cycles per wave-instruction per SIMD for fma / pk_fma / add,
// as a function of waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
  float a[8];
  v2f p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 1.f}; }
  const float c = 1.0001f, d = 0.5f;
  const v2f c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);
        if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], c2, d2);
        if (MODE == 2) a[i] = a[i] + c;
        if (MODE == 3) a[i] = a[i] * c;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 1024 * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_mul_f32"};
  printf("%-14s %-10s %10s %14s\n", "op", "waves/SIMD", "ms", "cyc/inst/SIMD@2.4GHz");
  for (int mode = 0; mode < 4; ++mode)
    for (int wps : {1, 2, 4, 8}) {
      dim3 grid(256 * wps), block(256);   // 4 waves per block -> wps blocks per CU
      float best = 1e9;
      for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double inst_per_simd = (double)iters * 32 * wps;  // each wave issues iters*32 instrs
      printf("%-14s %-10d %10.3f %14.2f\n", names[mode], wps, best, best * 1e-3 * 2.4e9 / inst_per_simd);
    }
  return 0;
}
