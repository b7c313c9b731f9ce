// Sum of the per-(half-)tile slabs of parameter-gradient partials that the training forward of embed_dim 64 leaves (fused_fwd32_tail.hpp,
// tail_bwd.hip) into the gradient tensors: argument block, slab format and the one-pass block role.  fused_aux.hip holds the two-pass
// kernels for large slab counts; fused_bwd.hip runs the role in the launch that sums the backward kernel's own slabs.
#pragma once
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

// One-pass sum as a role of 256 threads: block blk owns 32 float4 columns (512 bytes of every slab row), its eight 32-lane row groups take
// the slabs t = group, group + 8, ... with sixteen independent loads in flight per thread (the slabs are an 18 MB stream at 65 536 rows: what
// the sum costs is round trips, so they are what the loop minimises), the groups' partial sums meet in LDS in group order and the block adds
// its 128 elements into the gradient tensors.  Fixed order from run to run.  part: 8 x 32 float4 of LDS.
constexpr int kTailRoleBlocks = (kTailF4 + 31) / 32;         // 69
__device__ __forceinline__ void tail_reduce_role(const TailReduceArgs& a, int blk, float4* __restrict__ part) {
  const int tid = threadIdx.x, cg = tid & 31, grp = tid >> 5;
  const int c4 = blk * 32 + cg;
  int nt = a.count[a.count_idx];
  if (nt > a.ntiles_cap) nt = a.ntiles_cap;
  if (a.n_slabs >= 0) nt = a.n_slabs;
  float4 s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#define TRR_ADD(S, V) do { S.x += V.x; S.y += V.y; S.z += V.z; S.w += V.w; } while (0)
  if (c4 < kTailF4) {
    const float4* base = reinterpret_cast<const float4*>(a.tslab) + c4;
    int t = grp;
    for (; t + 8 * 15 < nt; t += 8 * 16) {
      float4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = base[(int64_t)(t + 8 * j) * kTailF4];
#pragma unroll
      for (int j = 0; j < 16; ++j) TRR_ADD(s[j & 7], v[j]);
    }
    for (; t + 8 * 3 < nt; t += 8 * 4) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = base[(int64_t)(t + 8 * j) * kTailF4];
#pragma unroll
      for (int j = 0; j < 4; ++j) TRR_ADD(s[j], v[j]);
    }
    for (; t < nt; t += 8) { const float4 v = base[(int64_t)t * kTailF4]; TRR_ADD(s[0], v); }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) TRR_ADD(s[j], s[j + 4]);
  part[grp * 32 + cg] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y), (s[0].z + s[1].z) + (s[2].z + s[3].z),
                                    (s[0].w + s[1].w) + (s[2].w + s[3].w));
#undef TRR_ADD
  __syncthreads();
  const int i = blk * 128 + tid;
  if (tid >= 128 || i > kTailVec + 9 * 64) return;
  const float* pf = reinterpret_cast<const float*>(part);
  float r = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) r += pf[g * 128 + tid];
  tail_slab_add(a, i, r);
}
// what launch_fused_bwd_merged needs to run the role: the slabs and where their sums go (fused_aux.hip fills the argument block)
void tail_reduce_args(const float* tslab, const Ragged& rg, int L, matcha_tensors& grads, bool halves, int n_slabs, bool rowmajor, TailReduceArgs& a);

}  // namespace matcha
