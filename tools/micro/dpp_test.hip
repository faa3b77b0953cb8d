#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float v = 100.f + l;
  out[0 * 64 + l] = dpp0<0x111, 0xf>(v);   // row_shr:1
  out[1 * 64 + l] = dpp0<0x118, 0xf>(v);   // row_shr:8
  out[2 * 64 + l] = dpp0<0x142, 0xa>(v);   // row_bcast15, rows 1,3
  out[3 * 64 + l] = dpp0<0x143, 0xc>(v);   // row_bcast31, rows 2,3
  out[4 * 64 + l] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.f), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));  // wave_shr:1
  out[5 * 64 + l] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  out[6 * 64 + l] = dpp0<0x101, 0xf>(v);   // row_shl:1
}
int main() {
  float* d; hipMalloc(&d, 7 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[7 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"row_shr1", "row_shr8", "bcast15(0xa)", "bcast31(0xc)", "wave_shr1", "readlane63", "row_shl1"};
  for (int r = 0; r < 7; ++r) { printf("%-13s", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
  return 0;
}
