// Shared device/host helpers for the audiotools_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AT_OK 0
#define AT_ERR_INVALID (-1)      // bad argument
#define AT_ERR_UNSUPPORTED (-2)  // valid request this entry point has no kernel for
#define AT_ERR_HIP(e) (-1000 - (int)(e))

#define AT_LAUNCH_CHECK()                              \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return AT_ERR_HIP(e__);     \
  } while (0)

namespace at {

// Wave-level rendezvous for LDS exchange between the lanes of ONE wave64.
// DS operations of a wave execute in issue order; the fences stop the
// compiler from moving LDS traffic across the exchange point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// wave64 sum via DPP-friendly xor shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Outer padding modes (torch.nn.functional.pad mode strings of the reference,
// audio_signal.py:1192-1194).
enum PadMode { PAD_REFLECT = 0, PAD_CONSTANT = 1, PAD_REPLICATE = 2, PAD_CIRCULAR = 3 };

// Map index v (relative to a length-T signal, may be out of range) through a
// padding mode.  Returns -1 for "constant zero".
__device__ __forceinline__ int64_t pad_index(int64_t v, int64_t T, int mode) {
  if (v >= 0 && v < T) return v;
  switch (mode) {
    case PAD_REFLECT: {
      if (T == 1) return 0;
      if (v < 0) v = -v;
      if (v >= T) v = 2 * (T - 1) - v;
      // one reflection is all torch allows (pad < T); clamp defensively
      return v < 0 ? 0 : (v >= T ? T - 1 : v);
    }
    case PAD_REPLICATE: return v < 0 ? 0 : T - 1;
    case PAD_CIRCULAR: { int64_t m = v % T; return m < 0 ? m + T : m; }
    default: return -1;
  }
}

}  // namespace at
