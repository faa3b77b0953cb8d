// Read-pattern microbenchmark for the inverse STFT (development aid): every wave reads whole
// spectrum rows (8200 B pitch) and writes one hop (2 KB) per frame; no transform.
// hipcc --offload-arch=gfx950 -O3 -o rdbench rdbench.hip && ./rdbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct P { const char* in; float* out; long frames; int run; int persistent; int prefetch; int spin; };

__device__ __forceinline__ void loadf(float2 (&d)[16], const char* in, long f, int lane) {
  const char* base = in + f * 8200L;
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = *reinterpret_cast<const float2*>(base + (i * 64 + lane) * 8);
}

__global__ __launch_bounds__(256) void rd(P p) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const long nruns = (p.frames + p.run - 1) / p.run;
  for (long r = wid; r < nruns; r += p.persistent ? nw : nruns) {
    float2 a[16], nx[16];
    long f = r * p.run;
    const long fe = f + p.run < p.frames ? f + p.run : p.frames;
    if (p.prefetch) loadf(nx, p.in, f, lane);
    for (; f < fe; ++f) {
      if (p.prefetch) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = nx[i];
        loadf(nx, p.in, f + 1 < fe ? f + 1 : f, lane);
      } else {
        loadf(a, p.in, f, lane);
      }
      float2 s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s[i] = make_float2(a[i].x + a[i + 4].x + a[i + 8].x + a[i + 12].x, a[i].y + a[i + 4].y + a[i + 8].y + a[i + 12].y);
      float v = s[0].x;
      for (int q = 0; q < p.spin; ++q) v = fmaf(v, 1.0001f, 0.5f);
      s[0].x = v;
      float2* o = reinterpret_cast<float2*>(p.out + f * 512);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[lane + 64 * i] = s[i];
    }
  }
}

int main() {
  const long frames = 882688;
  char* in; float* out;
  CK(hipMalloc(&in, frames * 8200L + 4096)); CK(hipMemset(in, 0, frames * 8200L + 4096));
  CK(hipMalloc(&out, frames * 512 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-5s %-5s %-5s %-5s %-6s %9s %9s\n", "run", "pers", "pref", "spin", "grid", "ms", "GB/s");
  struct C { int run, pers, pref, spin, gridmul; };
  std::vector<C> cs;
  for (int rep = 0; rep < 2; ++rep)
    for (int pref : {0, 1}) {
      cs.push_back({1, 0, 0, 0, 0});
      cs.push_back({16, 0, pref, 0, 0});
      cs.push_back({16, 1, pref, 0, 2});
      cs.push_back({16, 1, pref, 0, 4});
      cs.push_back({32, 1, pref, 0, 2});
      cs.push_back({16, 1, pref, 400, 2});
    }
  for (auto c : cs) {
    P p{in, out, frames, c.run, c.pers, c.pref, c.spin};
    long nruns = (frames + c.run - 1) / c.run;
    long grid = c.pers ? 256L * c.gridmul : (nruns + 3) / 4;
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(rd, dim3((unsigned)grid), dim3(256), 0, 0, p);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it && ms < best) best = ms;
    }
    double bytes = (double)frames * (8192 + 2048);
    printf("%-5d %-5d %-5d %-5d %-6ld %9.3f %9.1f\n", c.run, c.pers, c.pref, c.spin, grid, best, bytes / best / 1e6);
  }
  return 0;
}
