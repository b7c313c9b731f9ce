// Mean of the per-hyperedge losses in a fixed order, as a block-level role: loss_reduce_kernel (token_kernels.hip) runs it with 1024 threads,
// tail_bwd64_kernel (tail_bwd.hip) in one extra block of its launch (large batches: the row losses are complete when the forward kernel in
// front of it has finished, and nothing in that launch depends on the mean).
#pragma once
#include "kernels.hpp"

namespace matcha {

// red: 16 floats of LDS.  The order of the additions is that of 1024 threads whatever NT is (a block of NT threads plays 1024 / NT of them
// each), so both launches give the same bits: virtual thread T sums rows T, T + 1024, ... (float4 each, eight partial sums), an xor tree
// inside each virtual wavefront, the sixteen wavefronts in order.
template <int NT>
__device__ __forceinline__ void loss_reduce_role(const float* __restrict__ row_loss, int64_t B, float* __restrict__ out, int zero_recon, float* __restrict__ red) {
  static_assert(1024 % NT == 0 && NT % 64 == 0, "NT");
  constexpr int V = 1024 / NT;
  const int64_t B4 = ((uintptr_t)row_loss % 16 == 0) ? B / 4 : 0;           // float4 part (the workspace buffer is 256-byte aligned)
  const float4* r4 = reinterpret_cast<const float4*>(row_loss);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int T = threadIdx.x + NT * v;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int64_t i = T;
    for (; i + 3072 < B4; i += 4096) {
      const float4 v0 = r4[i], v1 = r4[i + 1024], v2 = r4[i + 2048], v3 = r4[i + 3072];
      a0 += v0.x + v0.y; a1 += v0.z + v0.w; a2 += v1.x + v1.y; a3 += v1.z + v1.w;
      a4 += v2.x + v2.y; a5 += v2.z + v2.w; a6 += v3.x + v3.y; a7 += v3.z + v3.w;
    }
    for (; i < B4; i += 1024) { const float4 u = r4[i]; a0 += u.x + u.y; a1 += u.z + u.w; }
    for (int64_t j = 4 * B4 + T; j < B; j += 1024) a2 += row_loss[j];
    float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[T >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t / (float)B;
    if (zero_recon) { out[1] = 0.f; out[2] = 0.f; }   // table front end: no reconstruction loss (instead of a memset in front of the forward)
  }
}

}  // namespace matcha
