// K15: torch.optim.AdamW(params, lr=1e-3) as main.py:630 / :671 builds it, fused over ONE flat buffer.
//
// HBM-streaming: 16 B read (p, g, m, v) + 16 B written (p, m, v, g := 0) per element, float4-vectorised.
// A "segment" is one tensor of the reference: per-tensor step counts and the grad-is-None skip are kept
// (SURVEY.md §7): segment s is updated only when touched[seg_group[s]] != 0, and only then does its step
// count advance.  Bias corrections are computed in double by a one-thread-per-segment prologue, like
// python does for torch (`1 - beta ** step`).
#include "common.hpp"

namespace matcha {

// coef[s] = {lr / (1 - b1^t), 1 / sqrt(1 - b2^t), active}
__global__ void adamw_prepare_kernel(int n_seg, const int32_t* __restrict__ seg_group, const int32_t* __restrict__ touched,
                                     int32_t* __restrict__ seg_step, float* __restrict__ coef, double lr, double b1, double b2) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  const bool active = touched ? (touched[seg_group ? seg_group[s] : 0] != 0) : true;
  float step_size = 0.f, inv_sqrt_bc2 = 0.f;
  if (active) {
    const int t = ++seg_step[s];
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    step_size = (float)(lr / bc1);
    inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
  coef[3 * s + 0] = step_size;
  coef[3 * s + 1] = inv_sqrt_bc2;
  coef[3 * s + 2] = active ? 1.f : 0.f;
}

__device__ __forceinline__ int find_segment(const int64_t* __restrict__ off, int n_seg, int64_t i) {
  int lo = 0, hi = n_seg;             // off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void adamw_elem(float& p, float& g, float& m, float& v, float step_size, float inv_sqrt_bc2,
                                           float decay, float omb1, float b2, float omb2, float eps, float gscale) {
  const float gg = g * gscale;
  p *= decay;                                         // p.mul_(1 - lr*wd)
  m = m + (gg - m) * omb1;                             // exp_avg.lerp_(grad, 1-beta1)
  v = v * b2 + omb2 * gg * gg;                         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;   // (sqrt(v)/sqrt(bc2)).add_(eps)
  p -= step_size * (m / denom);                        // p.addcdiv_(exp_avg, denom, value=-step_size)
  g = 0.f;                                             // opt.zero_grad() of the next step (main.py:175-176)
}

// each thread owns 4 consecutive elements; segment offsets must be multiples of 4 for the vector path
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ P, float* __restrict__ G, float* __restrict__ M,
                                                    float* __restrict__ V, int64_t n, const int64_t* __restrict__ seg_off,
                                                    int n_seg, const float* __restrict__ coef, float decay, float omb1,
                                                    float b2, float omb2, float eps, float gscale) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const int s = find_segment(seg_off, n_seg, i);
  const bool whole = (i + 3 < n) && (i + 3 < seg_off[s + 1]);
  if (whole) {
    if (coef[3 * s + 2] == 0.f) return;
    const float ss = coef[3 * s], ib = coef[3 * s + 1];
    float4 p = *reinterpret_cast<float4*>(P + i), g = *reinterpret_cast<float4*>(G + i);
    float4 m = *reinterpret_cast<float4*>(M + i), v = *reinterpret_cast<float4*>(V + i);
    adamw_elem(p.x, g.x, m.x, v.x, ss, ib, decay, omb1, b2, omb2, eps, gscale);
    adamw_elem(p.y, g.y, m.y, v.y, ss, ib, decay, omb1, b2, omb2, eps, gscale);
    adamw_elem(p.z, g.z, m.z, v.z, ss, ib, decay, omb1, b2, omb2, eps, gscale);
    adamw_elem(p.w, g.w, m.w, v.w, ss, ib, decay, omb1, b2, omb2, eps, gscale);
    *reinterpret_cast<float4*>(P + i) = p; *reinterpret_cast<float4*>(G + i) = g;
    *reinterpret_cast<float4*>(M + i) = m; *reinterpret_cast<float4*>(V + i) = v;
  } else {
    for (int e = 0; e < 4 && i + e < n; ++e) {
      const int64_t k = i + e;
      const int se = find_segment(seg_off, n_seg, k);
      if (coef[3 * se + 2] == 0.f) continue;
      adamw_elem(P[k], G[k], M[k], V[k], coef[3 * se], coef[3 * se + 1], decay, omb1, b2, omb2, eps, gscale);
    }
  }
}

}  // namespace matcha

using namespace matcha;

extern "C" int matcha_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const int64_t* seg_off, int32_t n_seg, const int32_t* seg_group, const int32_t* touched,
                                 int32_t* seg_step, float* seg_coef, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, double grad_scale, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && seg_off && seg_step && seg_coef, "matcha_adamw_step: null pointer");
  MATCHA_CHECK_ARG(n >= 0 && n_seg >= 1, "matcha_adamw_step: bad sizes n=%lld n_seg=%d", (long long)n, n_seg);
  MATCHA_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                   "matcha_adamw_step: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3((unsigned)cdiv(n_seg, 64)), dim3(64), 0, st, n_seg, seg_group, touched, seg_step,
                     seg_coef, lr, beta1, beta2);
  MATCHA_CHECK_LAUNCH("adamw_prepare_kernel");
  if (n == 0) return MATCHA_OK;
  // python-double scalars rounded to f32 once, as torch does for `1 - lr*wd`, `1 - beta1`, `1 - beta2`
  const float decay = (float)(1.0 - lr * weight_decay);
  ProfScope ps(MATCHA_PROF_ADAMW, 28.0 * (double)n, st);   // p, g, m, v read + p, m, v written (the kernel also writes g := 0: 32 B moved)
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)cdiv(cdiv(n, 4), 256)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n,
                     seg_off, n_seg, seg_coef, decay, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)grad_scale);
  MATCHA_CHECK_LAUNCH("adamw_kernel");
  return MATCHA_OK;
}
