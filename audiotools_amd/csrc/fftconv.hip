// Circular FFT convolution (rocFFT R2C / C2R + fused spectrum product) for gfx950.
//
// Replaces reference audiotools/core/effects.py:102-121 (EffectMixin.convolve):
//     rfft(ir, T), rfft(x, T), product, irfft(., T), scale by 1/clamp(max|ir|, 1e-5)
// -- a CIRCULAR convolution of length T (the reverb tail wraps to the head), kept as is.
// The reference's two extra FFTs of a unit impulse ("delta") are an algebraic identity and
// are not computed: the scale comes straight from the rolled IR.
//
// rocFFT does the inner DFTs (any length; T = 240000 = 2^7 3 5^4 at cfg4 is radix-2/3/5).
// Plans are cached per (device, T, rows) in a bounded LRU and own no device memory: rocFFT's work
// buffer is part of the caller's workspace.  The 1/T of the unnormalised inverse and the per-item scale are
// folded into the spectrum product, one pass over X.
#include "at_common.h"

#include <rocfft/rocfft.h>

#include <map>
#include <memory>
#include <mutex>
#include <tuple>

namespace {

// A cached rocFFT plan.  It owns NO device memory: rocFFT's work buffer is carved out of the
// caller's (torch-owned) workspace on every call, and the execution info object is per call, so
// two streams can run the same plan concurrently.  The cache is a small LRU: transforms under a
// partial mask change rows = B_masked * C from batch to batch and would otherwise grow it forever.
struct Plan {
  rocfft_plan plan = nullptr;
  size_t work_bytes = 0;
  uint64_t last_use = 0;
  ~Plan() { if (plan) rocfft_plan_destroy(plan); }
};

constexpr size_t MAX_PLANS = 24;
std::mutex g_mu;
bool g_setup = false;
uint64_t g_tick = 0;
std::map<std::tuple<int, int, int64_t, int64_t>, std::shared_ptr<Plan>> g_plans;  // (device, inverse, T, rows)

int get_plan(bool inverse, int64_t T, int64_t rows, std::shared_ptr<Plan>* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return AT_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_setup) {
    if (rocfft_setup() != rocfft_status_success) return AT_ERR_INVALID;
    g_setup = true;
  }
  auto key = std::make_tuple(dev, inverse ? 1 : 0, T, rows);
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    auto p = std::make_shared<Plan>();
    const size_t len[1] = {(size_t)T};
    rocfft_plan_description desc = nullptr;
    if (rocfft_plan_description_create(&desc) != rocfft_status_success) return AT_ERR_INVALID;
    const size_t F = (size_t)T / 2 + 1;
    const size_t rstride[1] = {1}, cstride[1] = {1};
    rocfft_status st;
    if (!inverse)
      st = rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                   nullptr, nullptr, 1, rstride, (size_t)T, 1, cstride, F);
    else
      st = rocfft_plan_description_set_data_layout(desc, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                   nullptr, nullptr, 1, cstride, F, 1, rstride, (size_t)T);
    if (st != rocfft_status_success) { rocfft_plan_description_destroy(desc); return AT_ERR_INVALID; }
    st = rocfft_plan_create(&p->plan, rocfft_placement_notinplace,
                            inverse ? rocfft_transform_type_real_inverse : rocfft_transform_type_real_forward,
                            rocfft_precision_single, 1, len, (size_t)rows, desc);
    rocfft_plan_description_destroy(desc);
    if (st != rocfft_status_success) { p->plan = nullptr; return AT_ERR_UNSUPPORTED; }
    if (rocfft_plan_get_work_buffer_size(p->plan, &p->work_bytes) != rocfft_status_success) return AT_ERR_INVALID;
    if (g_plans.size() >= MAX_PLANS) {   // evict the least recently used (in-flight users hold a shared_ptr)
      auto victim = g_plans.begin();
      for (auto jt = g_plans.begin(); jt != g_plans.end(); ++jt)
        if (jt->second->last_use < victim->second->last_use) victim = jt;
      g_plans.erase(victim);
    }
    it = g_plans.emplace(key, p).first;
  }
  it->second->last_use = ++g_tick;
  *out = it->second;
  return AT_OK;
}

int run(const Plan& p, void* in, void* out, void* work, size_t work_bytes, hipStream_t stream) {
  rocfft_execution_info info = nullptr;
  if (rocfft_execution_info_create(&info) != rocfft_status_success) return AT_ERR_INVALID;
  int rc = AT_OK;
  if (rocfft_execution_info_set_stream(info, stream) != rocfft_status_success) rc = AT_ERR_INVALID;
  if (rc == AT_OK && p.work_bytes) {
    if (!work || work_bytes < p.work_bytes ||
        rocfft_execution_info_set_work_buffer(info, work, p.work_bytes) != rocfft_status_success)
      rc = AT_ERR_INVALID;
  }
  if (rc == AT_OK) {
    void* ins[1] = {in};
    void* outs[1] = {out};
    if (rocfft_execute(p.plan, ins, outs, info) != rocfft_status_success) rc = AT_ERR_INVALID;
  }
  rocfft_execution_info_destroy(info);
  return rc;
}

inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

// X[b,c,f] *= H[b, c or 0, f] * scale[b, c or 0] / T
__global__ __launch_bounds__(256) void spectrum_product(float2* __restrict__ X, const float2* __restrict__ H,
                                                        const float* __restrict__ scale, int64_t B, int C, int Cir,
                                                        int64_t F, float inv_T) {
  for (int64_t row = blockIdx.y; row < B * C; row += gridDim.y) {  // row = b * C + c
    const int64_t b = row / C;
    const int c = (int)(row % C);
    const int64_t hrow = b * Cir + (Cir == 1 ? 0 : c);
    const float s = (scale ? scale[hrow] : 1.0f) * inv_T;
    float2* x = X + row * F;
    const float2* h = H + hrow * F;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < F; f += (int64_t)gridDim.x * blockDim.x) {
      const float2 a = x[f], w = h[f];
      x[f] = make_float2((a.x * w.x - a.y * w.y) * s, (a.x * w.y + a.y * w.x) * s);
    }
  }
}

}  // namespace

extern "C" {

// bytes of scratch: spectra (B*C + B*Cir) * (T/2+1) complex64 + the largest rocFFT work buffer of
// the three transforms (they run back to back on one stream and share it)
int64_t at_fftconv_workspace_bytes(int64_t B, int64_t C, int64_t Cir, int64_t T) {
  if (B < 0 || C <= 0 || Cir <= 0 || T <= 0) return AT_ERR_INVALID;
  if (B == 0) return 0;
  std::shared_ptr<Plan> fx, fh, inv;
  int rc;
  if ((rc = get_plan(false, T, B * C, &fx)) != AT_OK) return rc;
  if ((rc = get_plan(false, T, B * Cir, &fh)) != AT_OK) return rc;
  if ((rc = get_plan(true, T, B * C, &inv)) != AT_OK) return rc;
  size_t work = fx->work_bytes;
  if (fh->work_bytes > work) work = fh->work_bytes;
  if (inv->work_bytes > work) work = inv->work_bytes;
  return align256((B * C + B * Cir) * (T / 2 + 1) * 8) + (int64_t)work;
}

// x (B,C,T), ir (B,Cir,T) with Cir == 1 or Cir == C, scale (B,Cir) or NULL, out (B,C,T).
// NOTE: rocFFT's real inverse overwrites its input, so the product spectrum lives in `workspace`.
int at_fftconv_circ_f32(const float* x, const float* ir, const float* scale, int64_t B, int64_t C, int64_t Cir,
                        int64_t T, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (B == 0) return AT_OK;  // empty batch: nothing to do (torch hands out null data pointers)
  if (!x || !ir || !out || B < 0 || C <= 0 || T <= 0 || (Cir != 1 && Cir != C)) return AT_ERR_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t F = T / 2 + 1;
  std::shared_ptr<Plan> fx, fh, inv;
  int rc;
  if ((rc = get_plan(false, T, B * C, &fx)) != AT_OK) return rc;
  if ((rc = get_plan(false, T, B * Cir, &fh)) != AT_OK) return rc;
  if ((rc = get_plan(true, T, B * C, &inv)) != AT_OK) return rc;
  const int64_t spec_bytes = align256((B * C + B * Cir) * F * 8);
  if (!workspace || workspace_bytes < spec_bytes) return AT_ERR_INVALID;
  float2* X = reinterpret_cast<float2*>(workspace);
  float2* H = X + B * C * F;
  void* work = reinterpret_cast<char*>(workspace) + spec_bytes;
  const size_t work_bytes = (size_t)(workspace_bytes - spec_bytes);
  if ((rc = run(*fx, const_cast<float*>(x), X, work, work_bytes, st)) != AT_OK) return rc;
  if ((rc = run(*fh, const_cast<float*>(ir), H, work, work_bytes, st)) != AT_OK) return rc;
  const int bx = (int)((F + 255) / 256 < 64 ? (F + 255) / 256 : 64);
  hipLaunchKernelGGL(spectrum_product, dim3(bx, (unsigned)(B * C < 65535 ? B * C : 65535)), dim3(256), 0, st, X, H, scale, B, (int)C, (int)Cir, F,
                     1.0f / (float)T);
  AT_LAUNCH_CHECK();
  return run(*inv, X, out, work, work_bytes, st);
}

}  // extern "C"
