// Per-step helpers of the fused embed_dim-64 kernels (round 4: what is left of fused_fwd.hip after the four-wave tile forward went --
// fused_fwd32.hip is the forward on every path and prepares its own weights):
//   tail_slab_reduce/finish  fixed-order sum of the per-half-tile slabs of parameter-gradient partials the training forward writes
//                            (pff_n1, the three LayerNorms of the tail, the classifier) into the gradient tensors
//   tail_slab_small_kernel   the same in one launch for small batches
#include <stdlib.h>

#include "kernels.hpp"

namespace matcha {

constexpr int kTailVec = 2 * 4096;                 // offset of the vectors inside a slab
constexpr int kTailSlab = 2 * 4096 + 10 * 64;      // dW1, dW0, {gp, bp, g1, b1, g2, b2, wc, pff1_b, pff0_b} x 64, then bc (+ padding)

// Sum the per-tile slabs of the training forward in a fixed order and accumulate into the gradient tensors: a pure stream (133 MB per
// 65 536-row step with the four-wave forward, twice that with one slab per half tile), so what matters is bytes in flight.  Pass 1:
// block (column block of 64 float4, split s of the tile range) -- eight wavefronts, each reading whole 1 KB rows of its tiles with
// four independent chains (64 KB in flight per CU; the one-pass kernel with 4-byte loads ran at 3.2 TB/s) -> partial[s].  Pass 2:
// the kTailSplits partials of every element in split order, un-permuted into the gradient tensors.
constexpr int kTailSplits = 8;
constexpr int kTailF4 = kTailSlab / 4;                       // 2208 float4 per slab
constexpr int kTailColBlocks = (kTailF4 + 63) / 64;          // 35
struct TailReduceArgs {
  const float* tslab; const int32_t* count; int L; int ntiles_cap; int count_idx;
  float* partial;     // [kTailSplits][kTailSlab]
  float* dst[12];     // pff1_w, pff0_w, gp, bp, g1, b1, g2, b2, wc, pff1_b, pff0_b, bc
  float4* zero_buf; int64_t zero_n4;     // tail_slab_small_kernel: a buffer to zero in the same launch (the backward kernel's d x_hat)
  int n_slabs; int rowmajor;             // >= 0: slab count given by the launcher (tail_bwd64_kernel: one per workgroup), matrices row-major
  int with_mats; uint32_t slot_mask;     // which parts of the slabs are summed: the two matrices, vector slot v (bit v)
};
// element i of a summed slab into the gradient tensors: the two weight-gradient matrices arrive in the MFMA accumulator layout
// [wave][lane][register] (fused_fwd32_tail.hpp), the vectors as they are
__device__ __forceinline__ void tail_slab_add(const TailReduceArgs& a, int i, float s) {
  if (i < 8192 && !a.with_mats) return;
  if (i >= 8192 && ((a.slot_mask >> ((i - kTailVec) >> 6)) & 1u) == 0) return;
  if (i < 8192 && a.rowmajor) {
    a.dst[i >> 12][i & 4095] += s;
  } else if (i < 8192) {
    const int e = i & 4095, wv = e >> 10, ln = (e >> 4) & 63, reg = e & 15;
    const int row = 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), col = 32 * (wv >> 1) + (ln & 31);
    a.dst[i >> 12][row * 64 + col] += s;
  } else {
    const int v = (i - kTailVec) >> 6, j = (i - kTailVec) & 63;
    a.dst[2 + v][j] += s;
  }
}
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
// Small batches (a few dozen half tiles: the reference's own 384-row step): ONE pass.  Block = column block of 64 float4; its eight
// wavefronts take the slabs t = wave, wave + 8, ... (four rows in flight), the eight partial sums meet in LDS in wavefront order and the
// block adds its 256 elements into the gradient tensors itself: one launch instead of two, and the d x_hat buffer of the backward kernel
// is zeroed on the side (a third launch saved).  Another summation order than the two-pass kernels (like every small-batch kernel:
// compared at 2e-6 in tests/test_hip_properties.py), fixed from run to run.
__global__ __launch_bounds__(512) void tail_slab_small_kernel(TailReduceArgs a) {
  __shared__ float4 part[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + lane;
  if (a.zero_buf)
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < a.zero_n4; i += (int64_t)gridDim.x * 512) a.zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int nt = a.count[a.count_idx];
  if (nt > a.ntiles_cap) nt = a.ntiles_cap;
  if (a.n_slabs >= 0) nt = a.n_slabs;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (c4 < kTailF4) {
    const float4* base = reinterpret_cast<const float4*>(a.tslab) + c4;
    int t = wave;
    for (; t + 24 < nt; t += 32) {
      const float4 v0 = base[(int64_t)t * kTailF4], v1 = base[(int64_t)(t + 8) * kTailF4], v2 = base[(int64_t)(t + 16) * kTailF4], v3 = base[(int64_t)(t + 24) * kTailF4];
      TSR_ADD(s0, v0); TSR_ADD(s1, v1); TSR_ADD(s2, v2); TSR_ADD(s3, v3);
    }
    for (; t < nt; t += 8) { const float4 v = base[(int64_t)t * kTailF4]; TSR_ADD(s0, v); }
  }
  part[wave][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x >= 256 || i > kTailVec + 9 * 64) return;
  const float* pf = reinterpret_cast<const float*>(&part[0][0]);
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) s += pf[w * 256 + threadIdx.x];
  tail_slab_add(a, i, s);
}

size_t fused_fold_floats() { return (size_t)3 * (MATCHA_N_HEAD * 64 * 64 + MATCHA_N_HEAD * 64); }

size_t fused_tail_slab_floats(int64_t B, int L) { return (size_t)(ragged_tiles_cap(B, L) + 2) * kTailSlab; }
size_t fused_tail_partial_floats() { return (size_t)kTailSplits * kTailSlab; }
size_t fused_qkv_floats(int64_t B, int L) {          // the training forward's record: r rows + probabilities per (half tile, head)
  return (size_t)(ragged_halves_cap(B, L) + 2) * MATCHA_N_HEAD * kImgRecH;
}

int launch_tail_reduce(const float* tslab, const Ragged& rg, int L, matcha_tensors& g_, hipStream_t st, bool halves, float* partial, bool small,
                       float* zero_buf, size_t zero_bytes, int n_slabs, bool rowmajor, bool with_mats, uint32_t slot_mask) {
  MATCHA_CHECK_ARG(zero_bytes % 16 == 0 && (uintptr_t)zero_buf % 16 == 0 && (small || !zero_buf), "tail_reduce: zero_buf");
  TailReduceArgs a;
  a.tslab = tslab; a.count = rg.count; a.L = L; a.ntiles_cap = halves ? rg.nhalves : rg.ntiles; a.count_idx = halves ? 3 : 2;
  a.partial = partial;
  a.zero_buf = reinterpret_cast<float4*>(zero_buf); a.zero_n4 = (int64_t)(zero_bytes / 16);
  a.n_slabs = n_slabs; a.rowmajor = rowmajor ? 1 : 0; a.with_mats = with_mats ? 1 : 0; a.slot_mask = slot_mask;
  float* dst[12] = {g_.pff1_w, g_.pff0_w, g_.pff_ln_g, g_.pff_ln_b, g_.ln1_g, g_.ln1_b, g_.ln2_g, g_.ln2_b, g_.cls_w, g_.pff1_b, g_.pff0_b, g_.cls_b};
  for (int i = 0; i < 12; ++i) a.dst[i] = dst[i];
  if (small) {
    hipLaunchKernelGGL(tail_slab_small_kernel, dim3(kTailColBlocks), dim3(512), 0, st, a);
    MATCHA_CHECK_LAUNCH("tail_slab_small_kernel");
    return MATCHA_OK;
  }
  hipLaunchKernelGGL(tail_slab_reduce_kernel, dim3(kTailColBlocks, kTailSplits), dim3(512), 0, st, a);
  MATCHA_CHECK_LAUNCH("tail_slab_reduce_kernel");
  hipLaunchKernelGGL(tail_slab_finish_kernel, dim3((unsigned)cdiv(kTailSlab, 256)), dim3(256), 0, st, a);
  MATCHA_CHECK_LAUNCH("tail_slab_finish_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
