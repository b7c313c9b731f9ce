// Batched per-head d x d x d products of the merged layer-wise path (embed_dim >= 128; model.hip merged_weights / merged_chain):
//   forward   B_h = W_k[h]^T W_q[h],  M_all[:, h] = Wfc1[:, h] W_v[h]                                  (2 products x 8 heads)
//   backward  dW_q[h] += W_k[h] dB_h,  dW_k[h] += W_q[h] dB_h^T,  dWfc1[:, h] += dM_h W_v[h]^T,  dW_v[h] += Wfc1[:, h]^T dM_h   (4 x 8)
// Each is 2 d^3 flops on L2-resident operands -- nothing for the MFMA pipe -- but as 48 separate GEMM launches of one to four workgroups
// they cost 25-36 us apiece: 1.2 ms of the 8.8 ms step at d = 128 and 1.7 ms of the 12 ms step at d = 256 (profiles/r03_d128 / r03_c5
// kernel stats).  Here ONE launch takes up to four products x 8 heads: workgroup (tile i, tile j, product * 8 + head) computes a 64 x 64
// output tile, out[i][j] (+)= sum_x A(i, x) B(x, j), with every operand described by (pointer, row stride, column stride, head stride),
// so transposed operands need no copy.  Four wavefronts, v_mfma_f32_16x16x4_f32 on 2 x 2 tiles of 16 x 16 per wavefront, contraction in
// chunks of 16 staged through LDS in k-major order (operand reads are conflict-free), the next chunk's global loads in flight during
// the MFMAs of this one.  fp32 throughout.
#include "kernels.hpp"

namespace matcha {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBk = 16;            // contraction chunk
constexpr int kBld = 64 + 4;       // LDS row stride (floats) of a k-major [16][64] operand chunk

struct BmmArgs {
  BmmProduct p[4];
  int n;        // products in this launch
  int d;        // matrix order (multiple of 64)
};

// one float4 of an operand chunk: `fast` = the index (0: the output index i / j, 1: the contraction index x) that is contiguous in memory.
// fast == 1: thread t covers (o = t >> 2, x = 4 (t & 3) .. + 3); fast == 0: (x = t >> 4, o = 4 (t & 15) .. + 3)
__device__ __forceinline__ float4 chunk_load(const float* __restrict__ base, int64_t o_stride, int64_t x_stride, int fast, int o0, int x0, int t) {
  if (fast) {
    const int o = t >> 2, x = 4 * (t & 3);
    return *reinterpret_cast<const float4*>(base + (int64_t)(o0 + o) * o_stride + (x0 + x));
  }
  const int x = t >> 4, o = 4 * (t & 15);
  return *reinterpret_cast<const float4*>(base + (int64_t)(x0 + x) * x_stride + (o0 + o));
}
__device__ __forceinline__ void chunk_store(float* __restrict__ S, const float4& v, int fast, int t) {
  if (fast) {
    const int o = t >> 2, x = 4 * (t & 3);
    S[(x + 0) * kBld + o] = v.x; S[(x + 1) * kBld + o] = v.y; S[(x + 2) * kBld + o] = v.z; S[(x + 3) * kBld + o] = v.w;
  } else {
    const int x = t >> 4, o = 4 * (t & 15);
    *reinterpret_cast<float4*>(&S[x * kBld + o]) = v;
  }
}

__global__ __launch_bounds__(256) void bmm_heads_kernel(BmmArgs a) {
  __shared__ __attribute__((aligned(16))) float As[kBk * kBld];
  __shared__ __attribute__((aligned(16))) float Bs[kBk * kBld];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c16 = lane & 15, kq = lane >> 4;
  const int wr = wave & 1, wc = wave >> 1;
  const int prod = blockIdx.z >> 3, head = blockIdx.z & 7;
  const BmmProduct q = a.p[prod];
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  const float* A = q.A + (int64_t)head * q.a_hs;
  const float* B = q.B + (int64_t)head * q.b_hs;
  // the contiguous index of each operand (one of its two strides is 1: checked by the launcher)
  const int a_fast = q.a_cs == 1, b_fast = q.b_rs == 1;            // A: x contiguous;  B: x contiguous
  f32x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 an = chunk_load(A, q.a_rs, q.a_cs, a_fast, i0, 0, t);
  float4 bn = chunk_load(B, q.b_cs, q.b_rs, b_fast, j0, 0, t);        // B's "output index" is the column j: its stride is b_cs
  for (int x0 = 0; x0 < a.d; x0 += kBk) {
    __syncthreads();                                 // the previous chunk's MFMAs are done with the tiles
    chunk_store(As, an, a_fast, t);
    chunk_store(Bs, bn, b_fast, t);
    __syncthreads();
    if (x0 + kBk < a.d) {
      an = chunk_load(A, q.a_rs, q.a_cs, a_fast, i0, x0 + kBk, t);
      bn = chunk_load(B, q.b_cs, q.b_rs, b_fast, j0, x0 + kBk, t);
    }
#pragma unroll
    for (int kk = 0; kk < kBk / 4; ++kk) {
      const float* ap = As + (4 * kk + kq) * kBld + 32 * wr + c16;
      const float* bp = Bs + (4 * kk + kq) * kBld + 32 * wc + c16;
      const float a0 = ap[0], a1 = ap[16], b0 = bp[0], b1 = bp[16];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // lane (c16, kq) holds rows 4 kq + reg, column c16 of each 16 x 16 tile
  float* C = q.C + (int64_t)head * q.c_hs;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float* dst = C + (int64_t)(i0 + 32 * wr + 16 * m + 4 * kq + reg) * q.c_rs + (j0 + 32 * wc + 16 * n + c16);
        *dst = q.accumulate ? *dst + acc[m][n][reg] : acc[m][n][reg];
      }
}

}  // namespace

bool bmm_heads_supported(int d) { return d >= 64 && d % 64 == 0; }

int launch_bmm_heads(const BmmProduct* prods, int n, int d, hipStream_t st) {
  MATCHA_CHECK_ARG(prods && n >= 1 && n <= 4 && bmm_heads_supported(d), "bmm_heads: n=%d d=%d", n, d);
  BmmArgs a;
  a.n = n; a.d = d;
  for (int i = 0; i < n; ++i) {
    const BmmProduct& q = prods[i];
    MATCHA_CHECK_ARG(q.A && q.B && q.C, "bmm_heads: null operand");
    MATCHA_CHECK_ARG((q.a_cs == 1 || q.a_rs == 1) && (q.b_cs == 1 || q.b_rs == 1), "bmm_heads: an operand has no unit stride");
    // float4 loads: the non-unit strides, the head strides and the base pointers keep 16-byte alignment
    MATCHA_CHECK_ARG((q.a_rs == 1 ? q.a_cs : q.a_rs) % 4 == 0 && (q.b_rs == 1 ? q.b_cs : q.b_rs) % 4 == 0 && q.a_hs % 4 == 0 && q.b_hs % 4 == 0 &&
                         ((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0,
                     "bmm_heads: operand not 16-byte aligned");
    a.p[i] = q;
  }
  for (int i = n; i < 4; ++i) a.p[i] = prods[0];
  ProfScope ps(MATCHA_PROF_GEMM_NN, 2.0 * (double)d * d * d * MATCHA_N_HEAD * n, st);
  hipLaunchKernelGGL(bmm_heads_kernel, dim3(d / 64, d / 64, n * MATCHA_N_HEAD), dim3(256), 0, st, a);
  MATCHA_CHECK_LAUNCH("bmm_heads_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
