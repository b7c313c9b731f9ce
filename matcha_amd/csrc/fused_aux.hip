// Per-step helpers of the fused embed_dim-64 kernels (round 4: what is left of fused_fwd.hip after the four-wave tile forward went --
// fused_fwd32.hip is the forward on every path and prepares its own weights):
//   tail_slab_reduce/finish  fixed-order sum of the per-half-tile slabs of parameter-gradient partials the training forward writes
//                            (pff_n1, the three LayerNorms of the tail, the classifier) into the gradient tensors, two passes (thousands of
//                            slabs: the forward with its tail's backward in-kernel at a large batch).  Small batches and the split tail
//                            (tail_bwd.hip: 512 slabs) take the one-pass role of tail_reduce.hpp inside fbm_reduce_kernel's launch
#include <stdlib.h>

#include "kernels.hpp"
#include "tail_reduce.hpp"

namespace matcha {

__global__ __launch_bounds__(512) void tail_slab_reduce_kernel(TailReduceArgs a) {
  __shared__ float4 part[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + lane, sp = blockIdx.y;
  int nt = a.count[a.count_idx];                       // tiles (or half tiles) planned by ragged.hip: every one of them wrote its slab
  if (nt > a.ntiles_cap) nt = a.ntiles_cap;
  if (a.n_slabs >= 0) nt = a.n_slabs;
  const int lo = (int)((int64_t)nt * sp / kTailSplits), hi = (int)((int64_t)nt * (sp + 1) / kTailSplits);
  if (!a.with_mats && (blockIdx.x + 1) * 64 <= 8192 / 4) return;     // vector slots only: the matrix columns hold nothing
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0, s6 = s0, s7 = s0;
#define TSR_ADD(S, V) do { S.x += V.x; S.y += V.y; S.z += V.z; S.w += V.w; } while (0)
  if (c4 < kTailF4) {
    const float4* base = reinterpret_cast<const float4*>(a.tslab) + c4;
    int t = lo + wave;
    for (; t + 56 < hi; t += 64) {                     // eight independent 1 KB rows in flight per wavefront (64 KB per CU)
      const float4 v0 = base[(int64_t)t * kTailF4], v1 = base[(int64_t)(t + 8) * kTailF4], v2 = base[(int64_t)(t + 16) * kTailF4], v3 = base[(int64_t)(t + 24) * kTailF4];
      const float4 v4 = base[(int64_t)(t + 32) * kTailF4], v5 = base[(int64_t)(t + 40) * kTailF4], v6 = base[(int64_t)(t + 48) * kTailF4], v7 = base[(int64_t)(t + 56) * kTailF4];
      TSR_ADD(s0, v0); TSR_ADD(s1, v1); TSR_ADD(s2, v2); TSR_ADD(s3, v3); TSR_ADD(s4, v4); TSR_ADD(s5, v5); TSR_ADD(s6, v6); TSR_ADD(s7, v7);
    }
    for (; t < hi; t += 8) { const float4 v = base[(int64_t)t * kTailF4]; TSR_ADD(s0, v); }
  }
  TSR_ADD(s0, s4); TSR_ADD(s1, s5); TSR_ADD(s2, s6); TSR_ADD(s3, s7);
  part[wave][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (wave == 0 && c4 < kTailF4) {
    float4 s = part[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) { const float4 v = part[w][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    reinterpret_cast<float4*>(a.partial)[sp * kTailF4 + c4] = s;
  }
}
__global__ __launch_bounds__(256) void tail_slab_finish_kernel(TailReduceArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > kTailVec + 9 * 64) return;
  float s = 0.f;
#pragma unroll
  for (int sp = 0; sp < kTailSplits; ++sp) s += a.partial[sp * kTailSlab + i];
  tail_slab_add(a, i, s);
}
size_t fused_fold_floats() { return (size_t)3 * (MATCHA_N_HEAD * 64 * 64 + MATCHA_N_HEAD * 64); }

size_t fused_tail_slab_floats(int64_t B, int L) { return (size_t)(ragged_tiles_cap(B, L) + 2) * kTailSlab; }
size_t fused_tail_partial_floats() { return (size_t)kTailSplits * kTailSlab; }
size_t fused_qkv_floats(int64_t B, int L) {          // the training forward's record: r rows + probabilities per (half tile, head)
  return (size_t)(ragged_halves_cap(B, L) + 2) * MATCHA_N_HEAD * kImgRecH;
}

void tail_reduce_args(const float* tslab, const Ragged& rg, int L, matcha_tensors& g_, bool halves, int n_slabs, bool rowmajor, TailReduceArgs& a) {
  a.tslab = tslab; a.count = rg.count; a.L = L; a.ntiles_cap = halves ? rg.nhalves : rg.ntiles; a.count_idx = halves ? 3 : 2;
  a.partial = nullptr;
  a.n_slabs = n_slabs; a.rowmajor = rowmajor ? 1 : 0; a.with_mats = 1; a.slot_mask = 0x3FF;
  float* dst[12] = {g_.pff1_w, g_.pff0_w, g_.pff_ln_g, g_.pff_ln_b, g_.ln1_g, g_.ln1_b, g_.ln2_g, g_.ln2_b, g_.cls_w, g_.pff1_b, g_.pff0_b, g_.cls_b};
  for (int i = 0; i < 12; ++i) a.dst[i] = dst[i];
}

int launch_tail_reduce(const float* tslab, const Ragged& rg, int L, matcha_tensors& g_, hipStream_t st, bool halves, float* partial) {
  TailReduceArgs a;
  tail_reduce_args(tslab, rg, L, g_, halves, -1, false, a);
  a.partial = partial;
  hipLaunchKernelGGL(tail_slab_reduce_kernel, dim3(kTailColBlocks, kTailSplits), dim3(512), 0, st, a);
  MATCHA_CHECK_LAUNCH("tail_slab_reduce_kernel");
  hipLaunchKernelGGL(tail_slab_finish_kernel, dim3((unsigned)cdiv(kTailSlab, 256)), dim3(256), 0, st, a);
  MATCHA_CHECK_LAUNCH("tail_slab_finish_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
