// load->store ordering microbenchmark: does issuing the NEXT frame's loads before the current
// frame's stores remove the in-order vmcnt wait on store acknowledgements?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
struct P { const float* in; char* out; long frames; long T; int nframes_row; int mode; int spin; int lmode; int nt; };

__device__ __forceinline__ void loadf(float2 (&d)[16], const float* in, long f, const P& p, int lane) {
  const long row = f / p.nframes_row; const int fi = (int)(f % p.nframes_row);
  long s0 = row * p.T + (long)fi * 512;      // hop 512, frame 2048 floats (clamped at the row end)
  if ((long)fi * 512 + 2048 > p.T) s0 = row * p.T + p.T - 2048;
  if (p.lmode == 2) s0 = (f % 64) * 2048;              // 512 KB buffer: always L2/MALL resident
  const float2* q = reinterpret_cast<const float2*>(in + s0);
  if (p.lmode == 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = make_float2((float)f, (float)lane);
  } else if (p.lmode == 1) {                           // only the 2 KB hop is new data: 4 loads
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = q[lane + 64 * (i & 3)];
  } else if (p.nt & 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(q + lane + 64 * i)); d[i] = make_float2(t.x, t.y); }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = q[lane + 64 * i];
  }
}
__device__ __forceinline__ void storef_nt(const float2 (&d)[16], char* out, long f, int lane, float add) {
  char* base = out + f * 8200L;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    { v2f t = {d[q].x + add, d[q].y}; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(base + (q * 64 + lane) * 8)); }
    { v2f t = {d[q + 8].x + add, d[q + 8].y}; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(base + 8192 - (q * 64 + lane) * 8)); }
  }
  if (lane == 0) *reinterpret_cast<float2*>(base + 4096) = d[0];
}
__device__ __forceinline__ void storef(const float2 (&d)[16], char* out, long f, int lane, float add) {
  char* base = out + f * 8200L;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    *reinterpret_cast<float2*>(base + (q * 64 + lane) * 8) = make_float2(d[q].x + add, d[q].y);
    *reinterpret_cast<float2*>(base + 8192 - (q * 64 + lane) * 8) = make_float2(d[q + 8].x + add, d[q + 8].y);
  }
  if (lane == 0) *reinterpret_cast<float2*>(base + 4096) = d[0];
}

__global__ __launch_bounds__(256) void k(P p) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  float2 a[16], nx[16];
  if (p.mode == 0) {            // load -> (spin) -> store, nothing in flight across iterations
    for (long f = wid; f < p.frames; f += nw) {
      loadf(a, p.in, f, p, lane);
      float v = a[0].x;
      for (int s = 0; s < p.spin; ++s) v = fmaf(v, 1.0001f, 0.5f);
      if (p.nt & 2) storef_nt(a, p.out, f, lane, v * 1e-30f); else storef(a, p.out, f, lane, v * 1e-30f);
    }
  } else {                      // next frame's loads are issued BEFORE this frame's stores
    long f = wid;
    if (f < p.frames) loadf(nx, p.in, f, p, lane);
    for (; f < p.frames; f += nw) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = nx[i];
      if (f + nw < p.frames) loadf(nx, p.in, f + nw, p, lane);
      float v = a[0].x;
      for (int s = 0; s < p.spin; ++s) v = fmaf(v, 1.0001f, 0.5f);
      storef(a, p.out, f, lane, v * 1e-30f);
    }
  }
}

int main() {
  const long rows = 1024, T = 441000, nfr = 862, frames = rows * nfr;
  float* in; char* out;
  CK(hipMalloc(&in, rows * T * 4)); CK(hipMemset(in, 0, rows * T * 4));
  CK(hipMalloc(&out, frames * 8200L + 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-5s %-5s %-5s %-6s %9s %9s\n", "lmode", "nt", "spin", "grid", "ms", "GB/s");
  for (int lmode : {0, 1, 3})
  for (int spin : {0})
    for (int nt : {0, 1, 2, 3})
      for (int grid : {512}) {
        int mode = 0;
        P p{in, out, frames, T, (int)nfr, mode, spin, lmode, nt};
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (it && ms < best) best = ms;
        }
        printf("%-5d %-5d %-5d %-6d %9.3f %9.1f\n", lmode, nt, spin, grid, best, (frames * 8200.0 + rows * T * 4.0) / best / 1e6);
      }
  return 0;
}
