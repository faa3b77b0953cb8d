// gfx950 v_permlane16_swap / v_permlane32_swap: the 4 x 4 transpose between a wave's four 16-lane rows and four registers that the v2 STFT kernel uses
// (csrc/stft.hip, HX).  NOTE: read the two results into plain unsigned variables -- `__builtin_bit_cast(float, r_[1])` on the builtin's vector result
// compiled to element 0 with ROCm 7.2's clang (all four outputs of the transpose were one register).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void swap16(float& u, float& v) {
  auto r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(v), false, false);
  const unsigned x0 = r_[0], x1 = r_[1];
  u = __uint_as_float(x0); v = __uint_as_float(x1);
}
__device__ __forceinline__ void swap32(float& u, float& v) {
  auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(v), false, false);
  const unsigned x0 = r_[0], x1 = r_[1];
  u = __uint_as_float(x0); v = __uint_as_float(x1);
}
__global__ void k(float* out) {
  const int t = threadIdx.x;
  float r0 = 0 + t, r1 = 100 + t, r2 = 200 + t, r3 = 300 + t;   // reg a at lane t: 100 a + t
  swap16(r0, r1); swap16(r2, r3); swap32(r0, r2); swap32(r1, r3);
  out[t] = r0; out[64 + t] = r1; out[128 + t] = r2; out[192 + t] = r3;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;   // expect: lane row a, reg q  ==  100 a + (16 q + b)
  for (int q = 0; q < 4; ++q) for (int t = 0; t < 64; ++t) { const int a = t >> 4, b = t & 15; if (h[64 * q + t] != 100.f * a + 16 * q + b) ++bad; }
  printf("transpose mismatches: %d\n", bad);
  return 0;
}
