// Shared device/host helpers for the audiotools_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

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

constexpr int MAX_DEVICES = 64;

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize
// raised once per (kernel, device).  Keyed by the kernel's host address and the CURRENT device, so
// one process can drive several GPUs (a per-process flag made the second device's launch fail).
inline int allow_big_lds(const void* fn) {
  struct Entry { std::atomic<const void*> fn; std::atomic<uint64_t> devmask; };
  static Entry table[256];
  static std::atomic<int> count{0};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return AT_ERR_INVALID;
  const uint64_t bit = 1ull << dev;
  const int n = count.load(std::memory_order_acquire);
  for (int i = 0; i < n; ++i)
    if (table[i].fn.load(std::memory_order_relaxed) == fn)
      if (table[i].devmask.load(std::memory_order_acquire) & bit) return AT_OK;
  std::lock_guard<std::mutex> g(mu);
  int slot = -1;
  const int n2 = count.load(std::memory_order_relaxed);
  for (int i = 0; i < n2; ++i)
    if (table[i].fn.load(std::memory_order_relaxed) == fn) slot = i;
  if (slot < 0) {
    if (n2 >= 256) return AT_ERR_UNSUPPORTED;
    slot = n2;
    table[slot].fn.store(fn, std::memory_order_relaxed);
    table[slot].devmask.store(0, std::memory_order_relaxed);
    count.store(n2 + 1, std::memory_order_release);
  }
  if (!(table[slot].devmask.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return AT_ERR_HIP(e);
    table[slot].devmask.fetch_or(bit, std::memory_order_release);
  }
  return AT_OK;
}

// Compute units of the CURRENT device (cached per device; one process may drive several GPUs).
inline int device_cu_count() {
  static std::atomic<int> cached[MAX_DEVICES];
  int dev = 0, cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  const int c = cached[dev].load(std::memory_order_relaxed);
  if (c > 0) return c;
  if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) return 256;
  cached[dev].store(cu, std::memory_order_relaxed);
  return cu;
}

// Development switches.  The SHIPPED library has none: kernel selection never depends on the caller's environment
// (env_int_once folds to its default, the branches behind it are dead code).  A build with -DAT_DEV_KNOBS=1 (what the
// A/B tools under tools/ load through AT_LIB_PATH; `python -m audiotools_amd._native --dev`) reads them from the
// environment, ONCE per process and never on the launch path:
//   static const int v = at::env_int_once("AT_...", default);
#ifndef AT_DEV_KNOBS
#define AT_DEV_KNOBS 0
#endif
inline int env_int_once(const char* name, int dflt) {
#if AT_DEV_KNOBS
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// Streaming global accesses with a compile-time cache policy: NT = true marks the access
// non-temporal (`nt`: the line is not kept in the vector L1 / is the first to leave L2).  Which
// kernels use it is a per-kernel measurement (profiles/r02_notes.md): the v2 STFT's spectrum stores
// and the fused inverse STFT gain 2-4 %; LUFS, the overlap-save FIR and the resampler do not (and a
// load wrapped in a function cost the LUFS kernel its prefetch: every tile load was followed by
// s_waitcnt vmcnt(0) -- check the ISA when touching a load that is issued ahead of its use).
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef v2f_t v2f_a4_t __attribute__((aligned(4)));     // dword-aligned pair
template <bool NT> __device__ __forceinline__ float ldg(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ float2 ldg2(const float2* p) {
  if constexpr (NT) { const v2f_t u = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(p)); return make_float2(u.x, u.y); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ float2 ldg2_a4(const float* p) {
  if constexpr (NT) { const v2f_a4_t u = __builtin_nontemporal_load(reinterpret_cast<const v2f_a4_t*>(p)); return make_float2(u.x, u.y); }
  else { const v2f_a4_t u = *reinterpret_cast<const v2f_a4_t*>(p); return make_float2(u.x, u.y); }
}
template <bool NT> __device__ __forceinline__ float4 ldg4(const float4* p) {
  if constexpr (NT) { const v4f_t u = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p)); return make_float4(u.x, u.y, u.z, u.w); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ void stg(float* p, float v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT> __device__ __forceinline__ void stg2(float2* p, float2 v) {        // 8-byte aligned
  const v2f_t u = {v.x, v.y};
  if constexpr (NT) __builtin_nontemporal_store(u, reinterpret_cast<v2f_t*>(p)); else *p = v;
}
template <bool NT> __device__ __forceinline__ void stg2_a4(float* p, float x, float y) {
  const v2f_a4_t u = {x, y};
  if constexpr (NT) __builtin_nontemporal_store(u, reinterpret_cast<v2f_a4_t*>(p)); else *reinterpret_cast<v2f_a4_t*>(p) = u;
}

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

// sample fetch with centre reflect padding (torch.stft center=True) applied on top of the outer
// padding F.pad(audio, (pad, pad + right_pad), mode).  s indexes the outer-padded signal of length
// T2 and may be out of range.
__device__ __forceinline__ float fetch_padded(const float* __restrict__ xr, int64_t s, int64_t T, int64_t T2, int pad,
                                              int pad_mode) {
  int64_t u = s;
  if (u < 0) u = -u;
  if (u >= T2) u = 2 * (T2 - 1) - u;
  if (u < 0) u = 0;
  int64_t v = pad_index(u - pad, T, pad_mode);
  return v < 0 ? 0.0f : xr[v];
}

}  // namespace at
