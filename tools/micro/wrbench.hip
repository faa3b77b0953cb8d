// Store-pattern microbenchmark for the STFT output layout (development aid).
// hipcc --offload-arch=gfx950 -O3 -o wrbench wrbench.hip && ./wrbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct P { char* out; long frames; int pitch; int run; int persistent; int twostream; int spin; };

// one wave = one frame at a time; W = 8 (float2) or 16 (float4) bytes per lane per store
template <int W>
__global__ __launch_bounds__(256) void wr(P p) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const long nruns = (p.frames + p.run - 1) / p.run;
  for (long r = wid; r < nruns; r += p.persistent ? nw : nruns) {
    for (int j = 0; j < p.run; ++j) {
      const long f = r * p.run + j;
      if (f >= p.frames) break;
      char* base = p.out + f * (long)p.pitch;
      float v = (float)f;
      for (int s = 0; s < p.spin; ++s) v = v * 1.0001f + 0.5f;  // fake compute between frames
      const int nseg = 8192 / (64 * W);  // store instructions per frame (8192 B payload)
      if (p.twostream == 2) {
#pragma unroll
        for (int q = 0; q < nseg / 2; ++q)
          *reinterpret_cast<float2*>(base + (q * 64 + lane) * 8) = make_float2(v, v);
#pragma unroll
        for (int q = nseg / 2 - 1; q >= 0; --q)
          *reinterpret_cast<float2*>(base + 8192 - (q * 64 + lane) * 8) = make_float2(v, v);
      } else if (p.twostream) {
#pragma unroll
        for (int q = 0; q < nseg / 2; ++q) {
          if (W == 8) {
            *reinterpret_cast<float2*>(base + (q * 64 + lane) * 8) = make_float2(v, v);
            *reinterpret_cast<float2*>(base + 8192 - (q * 64 + lane) * 8) = make_float2(v, v);
          } else {
            *reinterpret_cast<float4*>(base + (q * 64 + lane) * 16) = make_float4(v, v, v, v);
            *reinterpret_cast<float4*>(base + 8192 - 16 - (q * 64 + lane) * 16 + 8) = make_float4(v, v, v, v);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < nseg; ++q) {
          if (W == 8) *reinterpret_cast<float2*>(base + (q * 64 + lane) * 8) = make_float2(v, v);
          else *reinterpret_cast<float4*>(base + (q * 64 + lane) * 16) = make_float4(v, v, v, v);
        }
      }
      if (lane == 0 && p.pitch > 8192) *reinterpret_cast<float2*>(base + 8192) = make_float2(v, v);
    }
  }
}

int main() {
  const long frames = 882688;
  char* buf;
  CK(hipMalloc(&buf, frames * 8208L + 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-6s %-6s %-5s %-5s %-4s %-5s %-6s %9s %9s\n", "W", "pitch", "run", "pers", "two", "spin", "grid", "ms", "GB/s");
  struct C { int W, pitch, run, pers, two, spin, gridmul; };
  std::vector<C> cs;
  for (int rep = 0; rep < 2; ++rep)
    for (int two : {0, 1, 2}) {
      cs.push_back({8, 8200, 1, 1, two, 0, 2});
      cs.push_back({8, 8200, 16, 0, two, 0, 0});
      cs.push_back({8, 8200, 16, 1, two, 0, 2});
    }
  for (auto c : cs) {
    P p{buf, frames, c.pitch, c.run, c.pers, c.two, c.spin};
    long nruns = (frames + c.run - 1) / c.run;
    long grid = c.pers ? 256L * c.gridmul : (nruns + 3) / 4;
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      if (c.W == 8) hipLaunchKernelGGL(wr<8>, dim3((unsigned)grid), dim3(256), 0, 0, p);
      else hipLaunchKernelGGL(wr<16>, dim3((unsigned)grid), dim3(256), 0, 0, p);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it && ms < best) best = ms;
    }
    double bytes = (double)frames * (c.pitch > 8192 ? 8200 : 8192);
    printf("%-6d %-6d %-5d %-5d %-4d %-5d %-6ld %9.3f %9.1f\n", c.W, c.pitch, c.run, c.pers, c.two, c.spin, grid, best, bytes / best / 1e6);
  }
  return 0;
}
