// Store-bandwidth microbenchmark: how much does the SEGMENT LENGTH of a store instruction cost?  (development aid)
// The fused STFT kernels write float2 bins; a wave owns 64 / L frames (L = n_fft / 32 lanes per frame), so one store
// instruction of the generic kernel writes, for each of its frames, L consecutive bins: segments of 8 L bytes at a frame pitch
// of (n_fft / 2 + 1) * 8 bytes -- 512 B at n_fft 2048, 128 B at n_fft 512, 64 B at n_fft 256.  Pattern B writes the same
// bytes as 512-byte runs of ONE frame per instruction.
// hipcc --offload-arch=gfx950 -O3 -o segbench segbench.hip && ./segbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct P { char* out; long groups; int M; int FW; int L; int mode; };

// a "group" = FW consecutive frames of M + 1 bins; persistent waves, runs of 16 groups
__global__ __launch_bounds__(256) void seg(P p) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const long pitch = (long)(p.M + 1) * 8;
  const int fs = lane / p.L, t = lane % p.L;
  for (long r = wid * 16; r < p.groups; r += nw * 16) {
    for (int j = 0; j < 16 && r + j < p.groups; ++j) {
      char* base = p.out + (r + j) * p.FW * pitch;
      const float v = (float)(r + j);
      if (p.mode == 0) {
        // shipped: instruction q writes bins t + L q of every frame of the group (16 instructions), + the Nyquist bins
#pragma unroll 4
        for (int q = 0; q < 16; ++q) *reinterpret_cast<float2*>(base + fs * pitch + (t + p.L * q) * 8) = make_float2(v, v);
        if (t == 0) *reinterpret_cast<float2*>(base + fs * pitch + p.M * 8) = make_float2(v, v);
      } else {
        // one frame per instruction: 64 consecutive bins (512 B); M / 64 instructions per frame (M >= 64), + Nyquist
        for (int f = 0; f < p.FW; ++f) {
          for (int c = 0; c < p.M / 64; ++c) *reinterpret_cast<float2*>(base + f * pitch + (64 * c + lane) * 8) = make_float2(v, v);
        }
        if (lane < p.FW) *reinterpret_cast<float2*>(base + lane * pitch + p.M * 8) = make_float2(v, v);
      }
    }
  }
}

// the resampler's output: 16 frames x `ph` phases of a tile; wave w writes 4 instructions of 4 x 64-byte pieces (mode 0)
// or the tile's contiguous bytes as 1 KB runs (mode 1)
__global__ __launch_bounds__(640) void rs(char* out, long tiles, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
    char* base = out + tl * 10240;
    const float v = (float)tl;
    if (mode == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(base + ((4 * (lane >> 4) + r) * 160 + 16 * wave + (lane & 15)) * 4) = v;
    } else {
      *reinterpret_cast<float4*>(base + (wave * 64 + lane) * 16) = make_float4(v, v, v, v);
    }
  }
}

int main() {
  char* buf;
  const long bytes_target = 2600000000L;
  CK(hipMalloc(&buf, bytes_target + (1 << 22)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-8s %-5s %-5s %-6s %9s %9s\n", "n_fft", "L", "mode", "seg B", "ms", "GB/s");
  for (int rep = 0; rep < 2; ++rep)
    for (int M : {1024, 512, 256, 128, 64}) {
      const int L = M / 16, FW = 64 / L;
      const long pitch = (long)(M + 1) * 8;
      const long groups = bytes_target / (pitch * FW);
      for (int mode : {0, 1}) {
        P p{buf, groups, M, FW, L, mode};
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(seg, dim3(512), dim3(256), 0, 0, p);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (it && ms < best) best = ms;
        }
        const double bytes = (double)groups * FW * pitch;
        printf("%-8d %-5d %-5d %-6d %9.3f %9.1f\n", 2 * M, L, mode, mode ? 512 : 8 * L, best, bytes / best / 1e6);
      }
    }
  const long tiles = 96256;
  for (int mode : {0, 1}) {
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(rs, dim3(512), dim3(640), 0, 0, buf, tiles, mode);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it && ms < best) best = ms;
    }
    printf("resampler tile stores, mode %d: %9.3f ms %9.1f GB/s\n", mode, best, tiles * 10240.0 / best / 1e6);
  }
  return 0;
}
