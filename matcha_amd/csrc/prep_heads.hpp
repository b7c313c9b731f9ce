// The per-step weight forms of the fused embed_dim-64 kernels (fused_fwd32.hip streams them, fused_bwd.hip reads the merged matrices and the
// folded projections): argument block, the fragment-stream layout and the block-level role function that builds them.
#pragma once
#include "bf16x3.hpp"
#include "kernels.hpp"

namespace matcha {

constexpr int kNMat = 20;                 // R_0 | R_{h+1} M_h (h = 0..6) | M_7 | conv0, conv1, conv1^T, conv0^T -- in consumption order
constexpr int kFragPerMat = 24;
constexpr int kFragU4 = kFragPerMat * 64;       // u32x4 per matrix
constexpr int kBiasR = 0, kBiasDyn = 8, kBiasConv0 = 9, kBiasConv1 = 10, kNBias = 11;    // rows of the f32 bias table behind the stream
constexpr int kPrepGridX = 4, kPrepGridY = 2, kPrepGridZ = MATCHA_N_HEAD + 1;
constexpr int kPrepBlocks = kPrepGridX * kPrepGridY * kPrepGridZ;
constexpr int kPrepLdsFloats = 16 * 65 + 64 * 68 + 512 + 64;

// ---- the per-step weight forms in ONE launch (rounds 1-3: fold_ln_kernel -> merge_heads_kernel -> fold_frag_kernel, 23 us of dependent
// latency in front of every forward; now 12) ------------------------------------------------------------------------------------------
//   fold:      W' = W * g, c = W . b for the three LayerNorm affines in front of Q / K / V (Modules.py:519-529); the fold happens on the way
//              into LDS and is written out for the backward's chain rule (fbm_chain_kernel reads W'q, W'k, W'v, cq, cv)
//   merge:     B_h = W'k^T W'q, b_h = W'k^T cq, M_h = Wfc1_h W'v, merged fc1 bias = fc1_b + Wfc1 cv   (16 products of 64^3 on MFMA tiles)
//   fragments: every matrix the forward streams, in MFMA-fragment order, in consumption order (R_0 | R_{h+1} M_h | M_7 | conv0 conv1
//              conv1^T conv0^T | one matrix of zeros): per 64 x 64 matrix 18 fragments of one float4 per lane, [wc][c = 0..7 | bias][lane],
//              lane (r, h) holds W[32 wc + r][8 c + 4 h .. + 3]; the bias fragment enters the accumulator as one more MFMA against 1
// grid (4 row slices of 16, 2 matrices, 8 heads + 1): block (slice, y, hd) computes 16 rows of B_hd (y = 0) or M_hd (y = 1) and writes them
// row-major (for the backward) and as fragments; slice 0 also computes the bias vector(s) it needs; z = 8: the conv fragments.
struct PrepArgs {
  const float* Wq; const float* Wk; const float* Wv;      // [512][64] as the reference holds them
  const float* gq; const float* gk; const float* gv; const float* bq; const float* bv;   // LayerNorm affines in front of them [64]
  const float* fc1_w; const float* fc1_b;
  const float* p0w; const float* p0b; const float* p1w; const float* p1b;
  float* fwq; float* fwk; float* fwv; float* fcq; float* fcv;          // folded forms (read by fbm_chain_kernel)
  float* B; float* M; float* bvec; float* bdyn;                        // merged forms (read by fused_bwdh_kernel / fbm_chain_kernel)
  u32x4* frag;                                                         // [kNMat + 1][kFragU4] bf16 planes, then the f32 bias table [kNBias][64]
};
// eight f32 values of one lane's contraction slots -> the three bf16 planes of fragment (c, wc) of matrix m, lane ln
__device__ __forceinline__ void frag_put8(u32x4* frag, int m, int c, int wc, int ln, const float* v8) {
  Frag3 b;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const P3 t = split2(v8[2 * q], v8[2 * q + 1]);
    b.h[q] = t.h; b.m[q] = t.m; b.l[q] = t.l;
  }
  u32x4* d = frag + (int64_t)m * kFragU4 + ((2 * c + wc) * 3) * 64 + ln;
  d[0] = b.h; d[64] = b.m; d[128] = b.l;
}
// One block of 256 threads = block (bx, by, bz) of the grid above; sm = kPrepLdsFloats floats of LDS (16-byte aligned).  A role function, so
// that the blocks can ride in another kernel's launch (front_fwd_kernel: the weight forms depend on the parameters only, the front end on the
// batch only -- 11 us of dependent latency off every step) as well as in prep_heads_kernel.
__device__ __forceinline__ void prep_heads_role(const PrepArgs& a, int bx, int by, int bz, float* __restrict__ sm) {
  float* As = sm;                                                      // [16][65]
  float* Bs = sm + 16 * 65;                                            // [64][68]
  float* cs = Bs + 64 * 68;                                            // [512] cq of this head (y = 0) / cv of this head or of all heads (y = 1, head 0)
  float* bs = cs + 512;                                                // [64] b_h or the merged fc1 bias
  const int slice = bx, hd = bz, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (hd == MATCHA_N_HEAD) {
    // conv0, conv1, conv1^T, conv0^T (matrices 16..19), one matrix of zeros behind the stream, and the two conv biases of the bias table
    const int id = bx + 4 * by;
    if (id > 4) return;
    for (int idx = tid; idx < 8 * 64; idx += 256) {       // (chunk c, block wc) x lane: one lane's eight slots, all three planes
      const int cw = idx >> 6, c = cw >> 1, wc = cw & 1, ln = idx & 63, r = ln & 31, hf = ln >> 5;
      const int n = 32 * wc + r;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * c + 8 * (j >> 2) + 4 * hf + (j & 3);
        v[j] = id == 0 ? a.p0w[n * 64 + k] : id == 1 ? a.p1w[n * 64 + k] : id == 2 ? a.p1w[k * 64 + n] : id == 3 ? a.p0w[k * 64 + n] : 0.f;
      }
      frag_put8(a.frag, 16 + id, c, wc, ln, v);
    }
    float* bias = reinterpret_cast<float*>(a.frag + (int64_t)(kNMat + 1) * kFragU4);
    if (id < 2 && tid < 64) bias[(kBiasConv0 + id) * 64 + tid] = id == 0 ? a.p0b[tid] : a.p1b[tid];
    return;
  }
  const bool isB = by == 0;
  const int64_t ho = (int64_t)hd * 4096;
  const float* Braw = (isB ? a.Wq : a.Wv) + ho;
  const float* gB = isB ? a.gq : a.gv;
  float* Bfold = (isB ? a.fwq : a.fwv) + ho;
  float* out = (isB ? a.B : a.M) + ho;
  const int mat = isB ? (hd == 0 ? 0 : 2 * hd - 1) : (hd < 7 ? 2 * hd + 2 : 15);     // position in the fragment stream
  {
    // the 16 x 64 left operand.  B_h: A(i, x) = W'k[x][16 slice + i] = Wk[x][16 slice + i] * gk[16 slice + i] (folded here, written out for
    // the backward); M_h: A(i, x) = Wfc1[16 slice + i][hd 64 + x]
    float av[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + 256 * t;
      if (isB) av[t] = a.Wk[ho + (int64_t)(i >> 4) * 64 + 16 * slice + (i & 15)] * a.gk[16 * slice + (i & 15)];
      else av[t] = a.fc1_w[(int64_t)(16 * slice + (i >> 6)) * 512 + hd * 64 + (i & 63)];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + 256 * t;
      As[isB ? (i & 15) * 65 + (i >> 4) : (i >> 6) * 65 + (i & 63)] = av[t];
      if (isB) a.fwk[ho + (int64_t)(i >> 4) * 64 + 16 * slice + (i & 15)] = av[t];
    }
  }
  {
    // the 64 x 64 right operand W'q_h / W'v_h = W * g (column scale), four float4 per thread in flight
    f32x4 bvv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bvv[u] = reinterpret_cast<const f32x4*>(Braw)[tid + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = (tid + 256 * u) * 4;
      const f32x4 gg = *reinterpret_cast<const f32x4*>(gB + (f & 63));
      f32x4 w = bvv[u];
      w.x *= gg.x; w.y *= gg.y; w.z *= gg.z; w.w *= gg.w;
      *reinterpret_cast<f32x4*>(&Bs[(f >> 6) * 68 + (f & 63)]) = w;
      if (slice == 0) reinterpret_cast<f32x4*>(Bfold)[tid + 256 * u] = w;
    }
  }
  __syncthreads();
  {
    const int c16 = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(As[c16 * 65 + 4 * kk + kq], Bs[(4 * kk + kq) * 68 + 16 * wave + c16], acc, 0, 0, 0);
    __syncthreads();                                  // every wavefront is done reading As: it now takes the 16 x 64 result tile
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int n = 16 * slice + 4 * kq + reg, k = 16 * wave + c16;
      out[n * 64 + k] = acc[reg];
      As[(4 * kq + reg) * 65 + k] = acc[reg];
    }
    __syncthreads();
    if (tid < 128) {
      // fragments: thread -> (row i of the tile, chunk c, lane half hf): eight contraction slots, three planes, 16-byte stores
      const int i = tid & 15, c = (tid >> 4) & 3, hf = tid >> 6;
      const int n = 16 * slice + i, wc = n >> 5, r = n & 31;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = As[i * 65 + 16 * c + 8 * (j >> 2) + 4 * hf + (j & 3)];
      frag_put8(a.frag, mat, c, wc, r + 32 * hf, v);
    }
  }
  if (slice != 0) return;
  // ---- slice 0 of every (matrix, head): the folded bias vector(s) it needs, the merged bias, the bias fragment -------------------------
  {
    // c[m] = sum_k W[m][k] b[k] (one xor tree over the 64 lanes per row).  y = 0: cq of this head;
    // y = 1: cv of this head -- of ALL heads for head 0, whose block builds the merged fc1 bias from them
    const float* Wraw = isB ? a.Wq : a.Wv;
    const float* bb = isB ? a.bq : a.bv;
    const int row0 = (!isB && hd == 0) ? 0 : hd * 64, nrow = (!isB && hd == 0) ? 512 : 64;
    const float bk = bb[lane];
    // 64 rows per wavefront and trip, lane = column: the 64 x 64 products are summed over the lanes by a TRANSPOSING butterfly -- at
    // offset o a lane keeps the rows whose index has bit o like its own lane id and hands the others to its partner -- 63 shuffles for 64
    // rows instead of 384, the same additions in the same order as group_sum<64> row by row (both partners of a butterfly step
    // compute the same sum), and lane l ends with row l
#pragma unroll 1
    for (int m0 = 64 * wave; m0 < nrow; m0 += 256) {
      float v[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) v[j] = Wraw[(int64_t)(row0 + m0 + j) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 64; ++j) v[j] *= bk;
#pragma unroll
      for (int half = 32; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
          const float keep = up ? v[j + half] : v[j];
          const float send = up ? v[j] : v[j + half];
          v[j] = keep + __shfl_xor(send, half, 64);
        }
      }
      cs[m0 + lane] = v[0];
    }
  }
  __syncthreads();
  if (tid < 64) (isB ? a.fcq : a.fcv)[hd * 64 + tid] = cs[tid];        // (head 0 of y = 1 holds all 512 but writes its own 64: the others write theirs)
  {
    const int o = tid >> 2, part = tid & 3;
    if (isB) {
      // b_h[o] = sum_m W'k[m][o] cq[m]
      float s_ = 0.f;
      for (int m = 16 * part; m < 16 * part + 16; ++m) s_ += (a.Wk[ho + m * 64 + o] * a.gk[o]) * cs[m];
      s_ += __shfl_xor(s_, 1, 64); s_ += __shfl_xor(s_, 2, 64);
      if (part == 0) { a.bvec[hd * 64 + o] = s_; bs[o] = s_; }
    } else if (hd == 0) {
      const float4* wrow = reinterpret_cast<const float4*>(a.fc1_w + o * 512 + 128 * part);
      const float4* cvp = reinterpret_cast<const float4*>(cs + 128 * part);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
      for (int m = 0; m < 32; m += 2) {
        const float4 w0 = wrow[m], c0 = cvp[m], w1 = wrow[m + 1], c1 = cvp[m + 1];
        s0 += (w0.x * c0.x + w0.y * c0.y) + (w0.z * c0.z + w0.w * c0.w);
        s1 += (w1.x * c1.x + w1.y * c1.y) + (w1.z * c1.z + w1.w * c1.w);
      }
      float s_ = s0 + s1;
      s_ += __shfl_xor(s_, 1, 64); s_ += __shfl_xor(s_, 2, 64);
      if (part == 0) { const float v = a.fc1_b[o] + s_; a.bdyn[o] = v; bs[o] = v; }
    } else if (part == 0) {
      bs[o] = 0.f;                                                     // the merged fc1 bias enters dyn once, with head 0
    }
  }
  __syncthreads();
  // the f32 bias table behind the stream: b_h (row hd) and, from head 0's M block, the merged fc1 bias (row kBiasDyn)
  if (tid < 64 && (isB || hd == 0))
    reinterpret_cast<float*>(a.frag + (int64_t)(kNMat + 1) * kFragU4)[(isB ? kBiasR + hd : kBiasDyn) * 64 + tid] = bs[tid];
}

// fills the argument block from the parameter tensors and the three workspace buffers (fused_fwd32.hip)
void prep_heads_args(const matcha_tensors& p, float* folded, float* merged, float* frag, PrepArgs& a);

}  // namespace matcha
