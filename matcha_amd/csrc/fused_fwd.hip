// Fused forward of the whole encoder + classifier tail for embed_dim 64 (the metric's dim):
//   X  ->  x_hat  ->  per head { Q,K,V = x_hat.W'^T + c  ->  attention (diag masked, pads attended)  ->  dyn += O.Wfc1_h^T }
//      ->  Y = mask*dropout(dyn + b)  ->  H1 = dropout(tanh(conv0 Y))  ->  H2 = conv1 H1 + Y
//      ->  LN_pff, LN1, LN2(X), (dn - sn)^2 . wc + bc  ->  per-hyperedge mean  ->  logit (+ weighted BCE term)
// (Modules.py:519-572, :353-376, :290-311; main.py:56).  Everything between X and the logits stays in LDS/registers.
// Three modes: inference (nothing saved); autograd forward (Y, H1, H2 saved, 768 B per token, for a later backward with an
// arbitrary dlogits); training step with the loss known here (opts.loss_in_forward): the kernel continues with
// dL/dlogit = alpha w (sigmoid(z) - y) / B through the tail and pff_n1 BACKWARD while Y, H1, H2 are still in LDS and emits
// d(dyn), dXs and per-tile partials of 12 parameter gradients -- nothing is saved.
//
// Work decomposition: one 256-thread workgroup per TILE of whole hyperedges, <= 63 real tokens + the shared padding token
// as the last row (its K/V rows come out of the same projection GEMMs); ragged.hip packs the tiles greedily (tile_meta).
// A 64x64x64 GEMM is split into four 32x32 quadrants, one per wave (32 f32 MFMAs each).
// The three LayerNorm affines in front of Q/K/V are folded into the projection weights once per step
// (W' = W * g, c = W . b), so one x_hat fragment set, held in registers for the whole tile, feeds all 24 projections.
#include <stdlib.h>

#include "kernels.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTM = 64;          // tile rows (tokens)
constexpr int kLdT = 68;         // LDS row stride (floats)
constexpr int kTileF = kTM * kLdT;
constexpr float kEps = 1e-5f;

// W'[n][k] = W[n][k] * g[k];  c[n] = sum_k W[n][k] * b[k]      (grid: (H*d rows, 3 matrices), 64 threads)
struct FoldArgs {
  const float* W[3]; const float* g[3]; const float* b[3];
  float* Wp[3]; float* c[3];
};
__global__ __launch_bounds__(64) void fold_ln_kernel(FoldArgs a) {
  const int z = blockIdx.y, n = blockIdx.x, k = threadIdx.x;
  const float w = a.W[z][n * 64 + k];
  a.Wp[z][n * 64 + k] = w * a.g[z][k];
  const float s = group_sum<64>(w * a.b[z][k]);
  if (k == 0) a.c[z][n] = s;
}

struct FusedFwdArgs {
  const float* X;                 // [Tn, 64]
  const int32_t* row_off;         // [B+1]
  const int32_t* tok_slot;        // [Tn]
  const int32_t* count;           // {Tr+1, Tr}
  const int32_t* tile_meta;       // [ntiles+2][4]
  const int32_t* tok_pos;         // [Tn]
  int64_t B;
  int L;
  const float* wq; const float* wk; const float* wv;     // folded, [512, 64]
  const float* cq; const float* ck; const float* cv;     // [512]
  const float* fc1_w; const float* fc1_b;                // [64, 512], [64]
  const float* p0w; const float* p0b; const float* p1w; const float* p1b;
  HeadParams hp;
  const float* y; const float* w;
  float* Y; float* H1; float* H2;                        // saved for backward (null: not saved)
  float* logits; float* row_loss;
  const uint64_t* seed;
  float p_fc1, p_pff;
  int dbg;                        // timing ablations only (MATCHA_FUSED_DBG): 1 = skip attention, 2 = skip projection GEMMs
  // training step with the loss known here (Trainer path): the kernel goes on with dL/dlogit = alpha w (sigmoid(z) - y) / B
  // and runs the backward of the classifier tail and of pff_n1 while Y, H1, H2 are still in LDS
  float* ddyn0;                   // [Tn, 64] gradient at the fc1 output (before bias; dropout / row mask applied); null: off
  float* dXs;                     // [Tn, 64] gradient into X through the static branch (layer_norm2)
  float* tslab;                   // [ntiles][kTailSlab] per-tile partials: dW1, dW0, 9 column-sum vectors, d bc
  float alpha_over_B;
  float* qkv;                     // [ntiles][8 heads][kImgRec]: register images of the Q, K, V tiles + attention probabilities for fused_bwd (null: not saved)
};
constexpr int kTailVec = 2 * 4096;                 // offset of the vectors inside a tile's slab
constexpr int kTailSlab = 2 * 4096 + 10 * 64;      // {gp, bp, g1, b1, g2, b2, wc, pff1_b, pff0_b} x 64, then bc (+ padding)

// LayerNorm backward of a row held as one float4 per lane over 16 lanes: dx = rstd (dxh - mean(dxh) - xh mean(dxh xh))
__device__ __forceinline__ float4 ln_bwd16(const float4& dxh, const float4& xh, float rstd) {
  const float a = group_sum16_dpp((dxh.x + dxh.y) + (dxh.z + dxh.w)) * (1.f / 64.f);
  const float b = group_sum16_dpp((dxh.x * xh.x + dxh.y * xh.y) + (dxh.z * xh.z + dxh.w * xh.w)) * (1.f / 64.f);
  return make_float4(rstd * (dxh.x - a - xh.x * b), rstd * (dxh.y - a - xh.y * b), rstd * (dxh.z - a - xh.z * b), rstd * (dxh.w - a - xh.w * b));
}
#define F4_FMA(acc, a, b) do { acc.x += (a).x * (b).x; acc.y += (a).y * (b).y; acc.z += (a).z * (b).z; acc.w += (a).w * (b).w; } while (0)
#define F4_ADD(acc, a) do { acc.x += (a).x; acc.y += (a).y; acc.z += (a).z; acc.w += (a).w; } while (0)
#define F4_MUL(a, b) make_float4((a).x * (b).x, (a).y * (b).y, (a).z * (b).z, (a).w * (b).w)

// stage a [64 x 64] fp32 block (row stride src_ld) into an LDS tile [64][68]
__device__ __forceinline__ void stage_tile(float* __restrict__ dst, const float* __restrict__ src, int64_t src_ld) {
  const int srow = threadIdx.x >> 4, sc4 = (threadIdx.x & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 16 * i;
    *reinterpret_cast<float4*>(&dst[row * kLdT + sc4]) = *reinterpret_cast<const float4*>(src + row * src_ld + sc4);
  }
}

// acc(32x32 quadrant) += A[rows 32*wr.., :] . B^T,  A fragments in registers, B tile in LDS as [n][k]
__device__ __forceinline__ f32x16 quad_gemm_regA(f32x16 acc, const float4 (&afr)[8], const float* __restrict__ Bs, int wc, int r, int h) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLdT + 8 * c + 4 * h]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[c].x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[c].y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[c].z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[c].w, b.w, acc, 0, 0, 0);
  }
  return acc;
}
// Projection with the TRANSPOSED product (weight rows = MFMA row operand, token rows = column operand): lane (r, h) ends
// with token row 32 wr + r and, in registers 4g..4g+3, the four consecutive features 32 wc + 8 g + 4 h + {0..3} -- one
// ds_write_b128 per group instead of four ds_write_b32, and the bias is the accumulator's initial value (no adds).
// VALU instructions are not free next to f32 MFMAs on this part (tools/ubench/mfma_valu.hip), so the epilogue matters.
// `gimg` (training): the same registers also go to global memory, 16 KB per tile in (wave quadrant, group, lane) order -- the
// backward kernel reloads them thread for thread instead of recomputing the projection (fused_bwd.hip).
__device__ __forceinline__ void proj_store_T(float* __restrict__ Ts, const float4 (&afr)[8], const float* __restrict__ Bs, const float* __restrict__ bias,
                                             int wr, int wc, int r, int h, bool skip, float* __restrict__ gimg) {
  f32x16 acc;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const float4 bv = *reinterpret_cast<const float4*>(&bias[32 * wc + 8 * gq + 4 * h]);
    acc[4 * gq] = bv.x; acc[4 * gq + 1] = bv.y; acc[4 * gq + 2] = bv.z; acc[4 * gq + 3] = bv.w;
  }
  if (!skip) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLdT + 8 * c + 4 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, afr[c].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, afr[c].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, afr[c].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, afr[c].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int gq = 0; gq < 4; ++gq)
    *reinterpret_cast<float4*>(&Ts[(32 * wr + r) * kLdT + 32 * wc + 8 * gq + 4 * h]) = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
  if (gimg) {
    f32x4* dst = reinterpret_cast<f32x4*>(gimg) + ((wr * 2 + wc) * 4) * 64 + 32 * h + r;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)          // written once, read ~1 ms later by another kernel: keep it out of the caches
      __builtin_nontemporal_store((f32x4){acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]}, dst + gq * 64);
  }
}
// same with the A tile in LDS ([row][k])
__device__ __forceinline__ f32x16 quad_gemm_ldsA(f32x16 acc, const float* __restrict__ As, const float* __restrict__ Bs, int wr, int wc, int r, int h) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLdT + 8 * c + 4 * h]);
    const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLdT + 8 * c + 4 * h]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}
// write a quadrant (+ per-column bias) to an LDS tile
__device__ __forceinline__ void quad_store(float* __restrict__ Ts, const f32x16& acc, const float* __restrict__ bias, int wr, int wc, int r, int h) {
  const int col = 32 * wc + r;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    Ts[row * kLdT + col] = acc[reg] + bv;
  }
}

// Attention output of ONE query token for one head from LDS tiles (8 lanes, lane `sub` owns features [8 sub, 8 sub + 8)):
// a tile's <= 63 tokens fill the workgroup's 32 lane groups in two passes, and the work per group is O(k), not O(k^2).
// Branch-free over the ML key slots (slots j >= k: clamped row, probability forced to 0).  O_i overwrites the token's own
// Q row (no other task reads it).
// Eight consecutive floats as four packed pairs: dot products and axpys compile to v_pk_mul / v_pk_fma.
typedef float f2 __attribute__((ext_vector_type(2)));
struct V8 { f2 a, b, c, d; };
__device__ __forceinline__ V8 ld8(const float* __restrict__ p) {
  const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
  V8 v;
  v.a = f2{x.x, x.y}; v.b = f2{x.z, x.w}; v.c = f2{y.x, y.y}; v.d = f2{y.z, y.w};
  return v;
}
__device__ __forceinline__ float dot8(const V8& u, const V8& v) {
  f2 s = u.a * v.a;
  s = __builtin_elementwise_fma(u.b, v.b, s);
  s = __builtin_elementwise_fma(u.c, v.c, s);
  s = __builtin_elementwise_fma(u.d, v.d, s);
  return s.x + s.y;
}
__device__ __forceinline__ void axpy8(V8& y, float w, const V8& x) {
  const f2 ww = {w, w};
  y.a = __builtin_elementwise_fma(ww, x.a, y.a); y.b = __builtin_elementwise_fma(ww, x.b, y.b);
  y.c = __builtin_elementwise_fma(ww, x.c, y.c); y.d = __builtin_elementwise_fma(ww, x.d, y.d);
}

template <int ML>
__device__ __forceinline__ void attn_row_fwd(float* __restrict__ Qs, const float* __restrict__ Ks, const float* __restrict__ Vs, int li, int li0, int k,
                                             int n_pad, int pad_row, int sub, float inv_temp, float* __restrict__ pimg) {
  const float padf = (float)n_pad;
  const bool hp = n_pad > 0;
  const int ii = li - li0;
  float p[ML], pp;
  int ro[ML];                                         // element offset of key / value row j (slots j >= k: clamped)
#pragma unroll
  for (int j = 0; j < ML; ++j) ro[j] = (li0 + (j < k ? j : 0)) * kLdT + 8 * sub;
  const V8 q = ld8(&Qs[li * kLdT + 8 * sub]);
  float mx = -3.4e38f;
#pragma unroll
  for (int j = 0; j < ML; ++j) {
    float a = group_sum8_dpp(dot8(q, ld8(&Ks[ro[j]]))) * inv_temp;
    a = (j == ii) ? -1e32f : a;                       // masked diagonal (Modules.py:443-445)
    p[j] = a;
    mx = (j < k) ? fmaxf(mx, a) : mx;
  }
  pp = group_sum8_dpp(dot8(q, ld8(&Ks[pad_row * kLdT + 8 * sub]))) * inv_temp;
  mx = hp ? fmaxf(mx, pp) : mx;
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < ML; ++j) {
    p[j] = (j < k) ? __expf(p[j] - mx) : 0.f;
    den += p[j];
  }
  pp = hp ? __expf(pp - mx) : 0.f;
  den += padf * pp;
  const float inv = __builtin_amdgcn_rcpf(den);
  V8 o;
  {
    const V8 vp = ld8(&Vs[pad_row * kLdT + 8 * sub]);
    const f2 w = {padf * pp * inv, padf * pp * inv};
    o.a = w * vp.a; o.b = w * vp.b; o.c = w * vp.c; o.d = w * vp.d;
  }
#pragma unroll
  for (int j = 0; j < ML; ++j) axpy8(o, p[j] * inv, ld8(&Vs[ro[j]]));
  *reinterpret_cast<float4*>(&Qs[li * kLdT + 8 * sub]) = make_float4(o.a.x, o.a.y, o.b.x, o.b.y);
  *reinterpret_cast<float4*>(&Qs[li * kLdT + 8 * sub + 4]) = make_float4(o.c.x, o.c.y, o.d.x, o.d.y);
  // training: row i of P for the backward pass -- slots 0..k-1 the real keys, slot 7 the per-slot padding probability (a
  // hyperedge with 8 real nodes has no padding slot, so the two never collide); the 8 lanes hold the same values
  if (pimg && sub == 0) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = j < ML ? p[j < ML ? j : 0] * inv : 0.f;
    if (hp) w[7] = pp * inv;
    f32x4* dst = reinterpret_cast<f32x4*>(pimg + li * 8);
    __builtin_nontemporal_store((f32x4){w[0], w[1], w[2], w[3]}, dst);
    __builtin_nontemporal_store((f32x4){w[4], w[5], w[6], w[7]}, dst + 1);
  }
}

// LayerNorm statistics of a 64-float row held as one float4 per lane over 16 lanes
__device__ __forceinline__ void ln_row16(const float4& v, float& mean, float& rstd) {
  const float s = group_sum16_dpp((v.x + v.y) + (v.z + v.w));
  mean = s * (1.f / 64.f);
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
  const float q = group_sum16_dpp((a * a + b * b) + (c * c + e * e));
  rstd = __builtin_amdgcn_rsqf(q * (1.f / 64.f) + kEps);   // v_rsq_f32 (1 ulp) instead of sqrt + divide
}
__device__ __forceinline__ float4 ln_apply(const float4& v, float mean, float rstd, const float4& g, const float4& b) {
  return make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y, (v.z - mean) * rstd * g.z + b.z, (v.w - mean) * rstd * g.w + b.w);
}

// 64 x 64 weight block: global -> registers (issued early), registers -> LDS tile (after the barrier that frees it).
// Named float4 registers + macros on purpose: with a struct passed by reference hipcc kept the prefetch registers in
// scratch memory and waited vmcnt(0) right after every load, i.e. no prefetch at all.
#define TILE_GLOAD(R, SRC, LD)                                                                           \
  do {                                                                                                   \
    const float* src__ = (SRC);                                                                          \
    R##0 = *reinterpret_cast<const float4*>(src__ + (int64_t)(srow) * (LD) + sc4);                       \
    R##1 = *reinterpret_cast<const float4*>(src__ + (int64_t)(srow + 16) * (LD) + sc4);                  \
    R##2 = *reinterpret_cast<const float4*>(src__ + (int64_t)(srow + 32) * (LD) + sc4);                  \
    R##3 = *reinterpret_cast<const float4*>(src__ + (int64_t)(srow + 48) * (LD) + sc4);                  \
  } while (0)
#define TILE_LSTORE(DST, R)                                                                              \
  do {                                                                                                   \
    *reinterpret_cast<float4*>(&(DST)[(srow) * kLdT + sc4]) = R##0;                                      \
    *reinterpret_cast<float4*>(&(DST)[(srow + 16) * kLdT + sc4]) = R##1;                                 \
    *reinterpret_cast<float4*>(&(DST)[(srow + 32) * kLdT + sc4]) = R##2;                                 \
    *reinterpret_cast<float4*>(&(DST)[(srow + 48) * kLdT + sc4]) = R##3;                                 \
  } while (0)

// LDS: one weight tile + Q/K/V tiles = 4 x 17 KiB -> two workgroups per CU; the next weight tile is always in flight
// in registers while the current GEMM / attention phase runs.
template <int ML>
__global__ __launch_bounds__(256, 2) void fused_fwd_kernel(FusedFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef FF_TIMING
  long long tph[24] = {0};
  long long tlast = wall_clock64();
#define FF_T(i) do { const long long now__ = wall_clock64(); tph[i] += now__ - tlast; tlast = now__; } while (0)
#else
#define FF_T(i) do { } while (0)
#endif
  float* Bs = lds;                    // current weight tile
  float* Qs = lds + 1 * kTileF;       // Q, then O (per head); later H1
  float* Ks = lds + 2 * kTileF;       // K; later H2
  float* Vs = lds + 3 * kTileF;       // V; later Y
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;          // quadrant of this wave
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;

  // ---- tile -> hyperedges [b0, b1), tokens [t0, t1) (+ the padding token as local row n_real) ----
  // window of first-token indices: a hyperedge starting at <= 63 - L ends at <= 63, so the tile holds <= 63 real tokens
  const int4 meta = reinterpret_cast<const int4*>(g.tile_meta)[blockIdx.x];   // planned by ragged.hip
  const int t0 = meta.x, n_real = meta.y, b0 = meta.z, n_h = meta.w;            // n_real <= 63
  if (n_h <= 0) return;                              // no hyperedge starts in this window
  const int tok_pad = g.count[1];                    // Tr: index of the shared padding token
  const float inv_temp = 0.125f;                     // 1/sqrt(64)

  // weight tiles are fetched ahead of their use into two alternating register sets (A: Wq / Wv, B: Wk / Wfc1)
  float4 wA0, wA1, wA2, wA3, wB0, wB1, wB2, wB3;
  TILE_GLOAD(wA, g.wq, 64);
  TILE_GLOAD(wB, g.wk, 64);
  // per-tile metadata in LDS: local row offsets of the tile's hyperedges and the folded projection biases of all heads
  // (small dependent global loads inside the head loop each cost a full L2 round trip at one or two waves per SIMD)
  int* roff = reinterpret_cast<int*>(lds + 4 * kTileF);          // [n_h + 1] (<= 64 hyperedges + 1)
  int* tinfo = roff + 80;                                         // [64] per token row: first row of its hyperedge | k << 8
  // dropout: keep <=> lowbias32(col ^ lowbias32(slot ^ key)) >= threshold; the inner hash depends on the token row only
  uint32_t* hrow1 = reinterpret_cast<uint32_t*>(tinfo + 64);       // [64] lowbias32(slot ^ key) for the fc1 dropout stream
  uint32_t* hrow2 = hrow1 + 64;                                    // [64] ... for the pff dropout stream
  float* cbias = lds + 4 * kTileF + 272;                          // [3][512]
  // the tail's parameter vectors, the three bias vectors of fc1 / conv0 / conv1 and the tile's labels and weights: fetched here once
  // (their latency hides behind the head loop) instead of as global loads at the head of each of the tail's short phases
  float* tpar = cbias + 3 * 512;                                  // [12][64]: gp bp g1 b1 g2 b2 wc | fc1_b p0b p1b | y w
  const bool lroff = n_h <= 78;                                   // more only when many all-padding rows share the window
  if (lroff)
    for (int i = tid; i <= n_h; i += 256) roff[i] = g.row_off[b0 + i] - t0;
  if (tid < n_real) {
    const int tp = g.tok_pos[t0 + tid];
    tinfo[tid] = (tid - (tp & 255)) | (tp & ~255);
  }
  const bool drop1 = g.p_fc1 > 0.f, drop2 = g.p_pff > 0.f;
  uint32_t thr1 = 0, thr2 = 0;
  float ks1 = 1.f, ks2 = 1.f;
  if (drop1) { thr1 = dropout_threshold(g.p_fc1); ks1 = 1.f / (1.f - g.p_fc1); }
  if (drop2) { thr2 = dropout_threshold(g.p_pff); ks2 = 1.f / (1.f - g.p_pff); }
  if (tid < 64 && (drop1 || drop2)) {
    const uint32_t slot = (uint32_t)g.tok_slot[tid < n_real ? t0 + tid : tok_pad];      // rows past the tokens: the padding token's slot
    hrow1[tid] = lowbias32(slot ^ rng_key(*g.seed, kStreamDropFc1));
    hrow2[tid] = lowbias32(slot ^ rng_key(*g.seed, kStreamDropPff));
  }
  for (int i = tid; i < 3 * 512; i += 256) cbias[i] = (i < 512) ? g.cq[i] : (i < 1024 ? g.ck[i - 512] : g.cv[i - 1024]);
  for (int i = tid; i < 10 * 64; i += 256) {
    const int v = i >> 6, j = i & 63;
    const float* src = v == 0 ? g.hp.gp : v == 1 ? g.hp.bp : v == 2 ? g.hp.g1 : v == 3 ? g.hp.b1 : v == 4 ? g.hp.g2 : v == 5 ? g.hp.b2
                     : v == 6 ? g.hp.wc : v == 7 ? g.fc1_b : v == 8 ? g.p0b : g.p1b;
    tpar[i] = src[j];
  }
  const bool lyw = g.row_loss && n_h <= 64;                       // labels / weights of the tile's hyperedges (training forward)
  if (lyw && tid < n_h) { tpar[10 * 64 + tid] = g.y[b0 + tid]; tpar[11 * 64 + tid] = g.w[b0 + tid]; }

  // ---- x_hat fragments straight from global memory: lane (r, h) holds k = 8c + 4h .. +3 of row 32 wr + r; the other
  //      half of the row lives in lane r + 32, so the LayerNorm statistics need one cross-half shuffle ----
  float4 afr[8];
  {
    const int row = 32 * wr + r;
    const bool valid = row <= n_real;
    const int64_t tok = row < n_real ? (int64_t)(t0 + row) : (int64_t)tok_pad;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      afr[c] = valid ? *reinterpret_cast<const float4*>(g.X + tok * 64 + 8 * c + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (afr[c].x + afr[c].y) + (afr[c].z + afr[c].w);
    }
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float a = afr[c].x - mean, b = afr[c].y - mean, e = afr[c].z - mean, f = afr[c].w - mean;
      q += (a * a + b * b) + (e * e + f * f);
    }
    q += __shfl_xor(q, 32, 64);
    const float rstd = __builtin_amdgcn_rsqf(q * (1.f / 64.f) + kEps);   // v_rsq_f32 (1 ulp) instead of sqrt + divide
#pragma unroll
    for (int c = 0; c < 8; ++c)
      afr[c] = make_float4((afr[c].x - mean) * rstd, (afr[c].y - mean) * rstd, (afr[c].z - mean) * rstd, (afr[c].w - mean) * rstd);
  }

  FF_T(0);
  f32x16 dyn = {0};
  for (int hd = 0; hd < MATCHA_N_HEAD; ++hd) {
    const int64_t wofs = (int64_t)hd * 64 * 64;
    const bool last = hd + 1 == MATCHA_N_HEAD;
    // ---- Q and K: W'q staged in Bs, W'k in the V tile (free until V is computed) -> one barrier pair for two GEMMs ----
    __syncthreads();                                   // everyone is done with Bs, Qs (previous fc1 GEMM) and Vs (attention)
    TILE_LSTORE(Bs, wA);
    TILE_LSTORE(Vs, wB);
    TILE_GLOAD(wA, g.wv + wofs, 64);
    TILE_GLOAD(wB, g.fc1_w + (int64_t)hd * 64, 512);    // fc1_w[n][hd*64 + k]: the head's column block as an [n][k] tile
    __syncthreads();
    FF_T(1);
    float* img = g.qkv ? g.qkv + ((int64_t)blockIdx.x * MATCHA_N_HEAD + hd) * kImgRec : nullptr;
    proj_store_T(Qs, afr, Bs, cbias + hd * 64, wr, wc, r, h, (g.dbg & 2) != 0, img);
    proj_store_T(Ks, afr, Vs, cbias + 512 + hd * 64, wr, wc, r, h, (g.dbg & 2) != 0, img ? img + 4096 : nullptr);
    // ---- V ----
    __syncthreads();                                   // both weight tiles consumed
    FF_T(2);
    TILE_LSTORE(Bs, wA);
    TILE_GLOAD(wA, last ? g.p0w : g.wq + wofs + 64 * 64, 64);
    __syncthreads();
    FF_T(3);
    proj_store_T(Vs, afr, Bs, cbias + 1024 + hd * 64, wr, wc, r, h, (g.dbg & 2) != 0, img ? img + 8192 : nullptr);
    __syncthreads();                                   // Q, K, V tiles complete; Bs free
    FF_T(4);
    TILE_LSTORE(Bs, wB);                               // fc1 block (read after the next barrier)
    TILE_GLOAD(wB, last ? g.p1w : g.wk + wofs + 64 * 64, 64);
    // ---- attention: 8 lanes per query token, two passes of 32 tokens ----
    if (!(g.dbg & 1)) {
      const int la = wave * 8 + (lane >> 3), lb = la + 32;
      float* pimg = img ? img + 3 * 4096 : nullptr;
      if (la < n_real) { const int ti = tinfo[la]; attn_row_fwd<ML>(Qs, Ks, Vs, la, ti & 255, ti >> 8, g.L - (ti >> 8), n_real, lane & 7, inv_temp, pimg); }
      if (lb < n_real) { const int ti = tinfo[lb]; attn_row_fwd<ML>(Qs, Ks, Vs, lb, ti & 255, ti >> 8, g.L - (ti >> 8), n_real, lane & 7, inv_temp, pimg); }
    }
    __syncthreads();
    FF_T(5);
    // ---- dyn += O_h . Wfc1[:, head block]^T ----
    if (!(g.dbg & 2)) dyn = quad_gemm_ldsA(dyn, Qs, Bs, wr, wc, r, h);
    FF_T(6);
  }
  __syncthreads();                                     // last fc1 GEMM done: Bs, Qs, Ks, Vs free
  FF_T(7);

  // the static branch's rows of X (LN2 in the tail): fetched now, in flight during the two pff GEMMs, instead of a dependent global
  // load per row in the middle of the tail's chain of short phases
  float4 xs0, xs1, xs2, xs3;
#define FF_XS_GLOAD(I)                                                                                   \
  do {                                                                                                   \
    const int row__ = srow + 16 * (I);                                                                   \
    const int64_t tok__ = row__ < n_real ? (int64_t)(t0 + row__) : (int64_t)tok_pad;                     \
    xs##I = *reinterpret_cast<const float4*>(g.X + tok__ * 64 + sc4);                                    \
  } while (0)
  FF_XS_GLOAD(0); FF_XS_GLOAD(1); FF_XS_GLOAD(2); FF_XS_GLOAD(3);
  // ---- Y = mask * dropout(dyn + b) -> Vs (+ global) ----
  uint32_t keep1 = 0, keep2 = 0;                        // this lane's 16 keep bits of the two dropout masks (reused by the backward part)
  float* Ys = Vs;
  float* H1s = Qs;
  float* H2s = Ks;
  TILE_LSTORE(Bs, wA);                                 // conv0 weight (fetched during the last head)
  {
    const int col = 32 * wc + r;
    const float bv = tpar[7 * 64 + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float v = 0.f;
      if (row < n_real) {                              // the padding token's row (and unused rows) are masked to 0
        v = dyn[reg] + bv;
        if (drop1) {
          const bool kp = lowbias32((uint32_t)col ^ hrow1[row]) >= thr1;
          keep1 |= kp ? (1u << reg) : 0u;
          v = kp ? v * ks1 : 0.f;
        }
        if (g.Y) g.Y[(int64_t)(t0 + row) * 64 + col] = v;
      } else if (row == n_real && g.Y) {
        g.Y[(int64_t)tok_pad * 64 + col] = 0.f;        // every tile writes the same zeros: benign
      }
      Ys[row * kLdT + col] = v;
    }
  }
  __syncthreads();
  FF_T(10);
  // ---- H1 = dropout(tanh(Y W0^T + b0)) -> Qs ----
  {
    f32x16 acc = {0};
    acc = quad_gemm_ldsA(acc, Ys, Bs, wr, wc, r, h);
    const int col = 32 * wc + r;
    const float bv = tpar[8 * 64 + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float v = fast_tanh(acc[reg] + bv);
      const int64_t tok = row < n_real ? (int64_t)(t0 + row) : (int64_t)tok_pad;
      if (drop2 && row <= n_real) {
        const bool kp = lowbias32((uint32_t)col ^ hrow2[row]) >= thr2;
        keep2 |= kp ? (1u << reg) : 0u;
        v = kp ? v * ks2 : 0.f;
      }
      if (g.H1 && row <= n_real) g.H1[tok * 64 + col] = v;
      H1s[row * kLdT + col] = v;
    }
  }
  __syncthreads();
  FF_T(11);
  TILE_LSTORE(Bs, wB);                                 // conv1 weight
  __syncthreads();
  // ---- H2 = H1 W1^T + b1 + Y -> Ks ----
  {
    f32x16 acc = {0};
    acc = quad_gemm_ldsA(acc, H1s, Bs, wr, wc, r, h);
    const int col = 32 * wc + r;
    const float bv = tpar[9 * 64 + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const float v = acc[reg] + bv + Ys[row * kLdT + col];
      const int64_t tok = row < n_real ? (int64_t)(t0 + row) : (int64_t)tok_pad;
      if (g.H2 && row <= n_real) g.H2[tok * 64 + col] = v;
      H2s[row * kLdT + col] = v;
    }
  }
  __syncthreads();
  FF_T(12);
  // ---- tail per token (16 lanes per row): out_t = sum_j (LN1(LN_pff(H2)) - LN2(X))_j^2 wc_j + bc ----
  float* outs = cbias;                                  // [64] per-token outputs (the folded biases are dead after the head loop)
  float* douts = cbias + 64;                            // [64] per-token gradient of them (training step)
  {
    const float4 Gp = *reinterpret_cast<const float4*>(tpar + 0 * 64 + sc4), Bp = *reinterpret_cast<const float4*>(tpar + 1 * 64 + sc4);
    const float4 G1 = *reinterpret_cast<const float4*>(tpar + 2 * 64 + sc4), B1 = *reinterpret_cast<const float4*>(tpar + 3 * 64 + sc4);
    const float4 G2 = *reinterpret_cast<const float4*>(tpar + 4 * 64 + sc4), B2 = *reinterpret_cast<const float4*>(tpar + 5 * 64 + sc4);
    const float4 Wc = *reinterpret_cast<const float4*>(tpar + 6 * 64 + sc4);
    const float bc = g.hp.bc[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      const float4 hv = *reinterpret_cast<const float4*>(&H2s[row * kLdT + sc4]);
      float m, rs;
      ln_row16(hv, m, rs);
      const float4 u = ln_apply(hv, m, rs, Gp, Bp);
      ln_row16(u, m, rs);
      const float4 dn = ln_apply(u, m, rs, G1, B1);
      const float4 xv = i == 0 ? xs0 : (i == 1 ? xs1 : (i == 2 ? xs2 : xs3));          // static branch: LN2 of the raw X row
      ln_row16(xv, m, rs);
      const float4 sn = ln_apply(xv, m, rs, G2, B2);
      const float a = dn.x - sn.x, b = dn.y - sn.y, c = dn.z - sn.z, e = dn.w - sn.w;
      const float o = group_sum16_dpp((a * a * Wc.x + b * b * Wc.y) + (c * c * Wc.z + e * e * Wc.w)) + bc;
      if ((tid & 15) == 0) outs[row] = o;
    }
  }
  __syncthreads();
  FF_T(13);
  // ---- per-hyperedge masked mean -> logit (+ BCE term) ----
  for (int e = tid; e < n_h; e += 256) {
    const int64_t b = b0 + e;
    const int li0 = lroff ? roff[e] : g.row_off[b0 + e] - t0;
    const int k = (lroff ? roff[e + 1] : g.row_off[b0 + e + 1] - t0) - li0;
    float tot = 0.f;
    for (int i = 0; i < k; ++i) tot += outs[li0 + i];
    const float z = tot / ((float)k + 1e-15f);
    g.logits[b] = z;
    const float yb = g.row_loss ? (lyw ? tpar[10 * 64 + e] : g.y[b]) : 0.f, wb = g.row_loss ? (lyw ? tpar[11 * 64 + e] : g.w[b]) : 0.f;
    if (g.row_loss) g.row_loss[b] = wb * (fmaxf(z, 0.f) - z * yb + log1pf(expf(-fabsf(z))));
    if (g.ddyn0) {                                     // main.py:56 backward: d bce / d z = w (sigmoid(z) - y) / B  (x alpha, main.py:166)
      const float dz = g.alpha_over_B * wb * (1.f / (1.f + expf(-z)) - yb);
      const float dout = dz / ((float)k + 1e-15f);
      for (int i = 0; i < k; ++i) douts[li0 + i] = dout;
    }
  }
  FF_T(8);
  if (!g.ddyn0) return;

  // =========== backward of the tail and of pff_n1 (Modules.py:290-311, :353-376), everything still in LDS ===========
  TILE_GLOAD(wA, g.p0w, 64);                           // conv0 weight for the last phase; Bs still holds conv1's
  __syncthreads();                                     // douts complete
  FF_T(14);
  float* tsl = g.tslab + (int64_t)blockIdx.x * kTailSlab;
  float4 aGp = make_float4(0.f, 0.f, 0.f, 0.f), aBp = aGp, aG1 = aGp, aB1 = aGp, aG2 = aGp, aB2 = aGp, aWc = aGp;
  float abc = 0.f;
  {
    const float4 Gp = *reinterpret_cast<const float4*>(tpar + 0 * 64 + sc4), Bp = *reinterpret_cast<const float4*>(tpar + 1 * 64 + sc4);
    const float4 G1 = *reinterpret_cast<const float4*>(tpar + 2 * 64 + sc4), B1 = *reinterpret_cast<const float4*>(tpar + 3 * 64 + sc4);
    const float4 G2 = *reinterpret_cast<const float4*>(tpar + 4 * 64 + sc4), B2 = *reinterpret_cast<const float4*>(tpar + 5 * 64 + sc4);
    const float4 Wc = *reinterpret_cast<const float4*>(tpar + 6 * 64 + sc4);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      const float dout = row < n_real ? douts[row] : 0.f;
      const float4 hv = *reinterpret_cast<const float4*>(&H2s[row * kLdT + sc4]);
      float m, rh, ru, rx;
      ln_row16(hv, m, rh);
      const float4 hh = ln_apply(hv, m, rh, one4, zero4);
      const float4 u = make_float4(hh.x * Gp.x + Bp.x, hh.y * Gp.y + Bp.y, hh.z * Gp.z + Bp.z, hh.w * Gp.w + Bp.w);
      ln_row16(u, m, ru);
      const float4 uh = ln_apply(u, m, ru, one4, zero4);
      const int64_t tok = row < n_real ? (int64_t)(t0 + row) : (int64_t)tok_pad;
      const float4 xv = i == 0 ? xs0 : (i == 1 ? xs1 : (i == 2 ? xs2 : xs3));          // the rows fetched before the pff GEMMs
      ln_row16(xv, m, rx);
      const float4 xh = ln_apply(xv, m, rx, one4, zero4);
      const float4 df = make_float4((uh.x * G1.x + B1.x) - (xh.x * G2.x + B2.x), (uh.y * G1.y + B1.y) - (xh.y * G2.y + B2.y),
                                    (uh.z * G1.z + B1.z) - (xh.z * G2.z + B2.z), (uh.w * G1.w + B1.w) - (xh.w * G2.w + B2.w));
      aWc.x += df.x * df.x * dout; aWc.y += df.y * df.y * dout; aWc.z += df.z * df.z * dout; aWc.w += df.w * df.w * dout;
      if ((tid & 15) == 0) abc += dout;
      const float4 ddn = make_float4(2.f * df.x * Wc.x * dout, 2.f * df.y * Wc.y * dout, 2.f * df.z * Wc.z * dout, 2.f * df.w * Wc.w * dout);
      const float4 dsn = make_float4(-ddn.x, -ddn.y, -ddn.z, -ddn.w);
      // layer_norm1 (dynamic branch), then pff_n1.layer_norm
      F4_FMA(aG1, ddn, uh); F4_ADD(aB1, ddn);
      const float4 du = ln_bwd16(F4_MUL(ddn, G1), uh, ru);
      F4_FMA(aGp, du, hh); F4_ADD(aBp, du);
      const float4 dh = ln_bwd16(F4_MUL(du, Gp), hh, rh);
      *reinterpret_cast<float4*>(&H2s[row * kLdT + sc4]) = dh;              // dH2 replaces H2 (zero rows past the tile's tokens)
      // layer_norm2 (static branch) -> gradient into X
      F4_FMA(aG2, dsn, xh); F4_ADD(aB2, dsn);
      const float4 dxs = ln_bwd16(F4_MUL(dsn, G2), xh, rx);
      if (row <= n_real) *reinterpret_cast<float4*>(g.dXs + tok * 64 + sc4) = dxs;   // the padding token's row is zero (dout = 0)
    }
  }
  __syncthreads();
  FF_T(15);
  // ---- conv1: dW1[n][k] += sum_t dH2[t][n] H1[t][k];  d b1 = column sums of dH2;  dZ1 = (dH2 W1) * dropmask * tanh' ----
  float* dZs = H1s;
  float cs1 = 0.f, cs0 = 0.f;
  {
    f32x16 aw = {0};
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const int t = 2 * m + h;
      const float gh = H2s[t * kLdT + 32 * wr + r];
      cs1 += gh;
      aw = __builtin_amdgcn_mfma_f32_32x32x2f32(gh, H1s[t * kLdT + 32 * wc + r], aw, 0, 0, 0);
    }
    // the slab is private scratch: it keeps the ACCUMULATOR layout ([wave][lane][16 registers] = four 16-byte stores per lane,
    // 4 KB contiguous per wave) and tail_slab_reduce_kernel un-permutes once; written row-major this was 16 four-byte stores per
    // lane and matrix, and store issue is what such an epilogue costs (gemm_wide.hip measured it)
    f32x4* ts4 = reinterpret_cast<f32x4*>(tsl) + (wave * 64 + lane) * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) ts4[gq] = (f32x4){aw[4 * gq], aw[4 * gq + 1], aw[4 * gq + 2], aw[4 * gq + 3]};
  }
  {
    f32x16 acc = {0};
#pragma unroll
    for (int c = 0; c < 8; ++c) {                      // dZ1[t][k] = sum_n dH2[t][n] W1[n][k]  (W1 stored [n][k]: column walk)
      const float4 a = *reinterpret_cast<const float4*>(&H2s[(32 * wr + r) * kLdT + 8 * c + 4 * h]);
      const float* wp = &Bs[(8 * c + 4 * h) * kLdT + 32 * wc + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[kLdT], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * kLdT], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * kLdT], acc, 0, 0, 0);
    }
    __syncthreads();                                   // every wave is done reading H1 (weight gradient) and W1
    FF_T(16);
    const int col = 32 * wc + r;
    const float unscale = drop2 ? 1.f - g.p_pff : 1.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const float hval = H1s[row * kLdT + col] * unscale;                   // tanh value (0 where dropped)
      float v = acc[reg];
      const int64_t tok = row < n_real ? (int64_t)(t0 + row) : (int64_t)tok_pad;
      if (drop2) v = ((keep2 >> reg) & 1u) ? v * ks2 : 0.f;        // the mask drawn in the forward part (rows past the tokens: v is 0)
      dZs[row * kLdT + col] = v * (1.f - hval * hval);
    }
    TILE_LSTORE(Bs, wA);                               // conv0 weight
  }
  __syncthreads();
  FF_T(17);
  // ---- conv0: dW0[n][k] += sum_t dZ1[t][n] Y[t][k];  d b0 = column sums of dZ1;  d dyn = (dZ1 W0 + dH2) * dropmask * rowmask ----
  {
    f32x16 aw = {0};
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const int t = 2 * m + h;
      const float gz = dZs[t * kLdT + 32 * wr + r];
      cs0 += gz;
      aw = __builtin_amdgcn_mfma_f32_32x32x2f32(gz, Ys[t * kLdT + 32 * wc + r], aw, 0, 0, 0);
    }
    f32x4* ts4 = reinterpret_cast<f32x4*>(tsl + 4096) + (wave * 64 + lane) * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) ts4[gq] = (f32x4){aw[4 * gq], aw[4 * gq + 1], aw[4 * gq + 2], aw[4 * gq + 3]};
  }
  {
    f32x16 acc = {0};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(&dZs[(32 * wr + r) * kLdT + 8 * c + 4 * h]);
      const float* wp = &Bs[(8 * c + 4 * h) * kLdT + 32 * wc + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[kLdT], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * kLdT], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * kLdT], acc, 0, 0, 0);
    }
    const int col = 32 * wc + r;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row <= n_real) {
        float v = 0.f;                                 // the padding token's row is masked (Modules.py:614)
        if (row < n_real) {
          v = acc[reg] + H2s[row * kLdT + col];        // residual: H2 = conv1(H1) + Y
          if (drop1) v = ((keep1 >> reg) & 1u) ? v * ks1 : 0.f;
        }
        H2s[row * kLdT + col] = v;                     // in place: in this phase only the owning lane reads H2s[row][col]
      }
    }
  }
  // ---- parameter-vector partials of this tile: rows of the 16 staging groups, then the two column sums ----
  __syncthreads();                                     // all reads of the tiles are done: Bs / dZs are scratch now
  FF_T(18);
  // d dyn leaves as whole 256-byte rows (16 lanes x 16 bytes) instead of 16 four-byte stores per lane in the accumulator layout
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 16 * i;
    if (row <= n_real) {
      const int64_t tok = row < n_real ? (int64_t)(t0 + row) : (int64_t)tok_pad;
      *reinterpret_cast<float4*>(g.ddyn0 + tok * 64 + sc4) = *reinterpret_cast<const float4*>(&H2s[row * kLdT + sc4]);
    }
  }
  // [16][7][64] + the column sums: 7312 floats laid over the weight tile and the front of the dZ tile (tiles 0, 1 -- both dead
  // after the barrier above).  NOT over tile 2: the rows of H2s are still being copied out by slower waves in this phase.
  float* red = Bs;
  *reinterpret_cast<float4*>(&red[(srow * 7 + 0) * 64 + sc4]) = aGp; *reinterpret_cast<float4*>(&red[(srow * 7 + 1) * 64 + sc4]) = aBp;
  *reinterpret_cast<float4*>(&red[(srow * 7 + 2) * 64 + sc4]) = aG1; *reinterpret_cast<float4*>(&red[(srow * 7 + 3) * 64 + sc4]) = aB1;
  *reinterpret_cast<float4*>(&red[(srow * 7 + 4) * 64 + sc4]) = aG2; *reinterpret_cast<float4*>(&red[(srow * 7 + 5) * 64 + sc4]) = aB2;
  *reinterpret_cast<float4*>(&red[(srow * 7 + 6) * 64 + sc4]) = aWc;
  float* redc = red + 16 * 7 * 64;                     // [2][64] column sums, [16] bc partials
  cs1 += __shfl_xor(cs1, 32, 64);
  cs0 += __shfl_xor(cs0, 32, 64);
  if (wc == 0 && h == 0) { redc[32 * wr + r] = cs1; redc[64 + 32 * wr + r] = cs0; }
  if ((tid & 15) == 0) redc[128 + srow] = abc;
  __syncthreads();
  FF_T(19);
  for (int i = tid; i < 7 * 64; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q * 7 * 64 + i];
    tsl[kTailVec + i] = t;
  }
  if (tid < 128) tsl[kTailVec + 7 * 64 + tid] = redc[tid];
  if (tid == 0) {
    float t = 0.f;
    for (int q = 0; q < 16; ++q) t += redc[128 + q];
    tsl[kTailVec + 9 * 64] = t;
  }
  FF_T(9);
#ifdef FF_TIMING
  if (blockIdx.x == 1000 && tid == 0)
    printf("fused_fwd wg1000 us: setup %.1f | 8 heads: stageQK %.1f gemmQK %.1f stageV %.1f gemmV %.1f attn %.1f fc1 %.1f | drain %.1f tail-fwd %.1f tail-bwd %.1f\n",
           tph[0] * 0.01, tph[1] * 0.01, tph[2] * 0.01, tph[3] * 0.01, tph[4] * 0.01, tph[5] * 0.01, tph[6] * 0.01, tph[7] * 0.01, tph[8] * 0.01, tph[9] * 0.01);
  if (blockIdx.x == 1000 && tid == 0)
    printf("   tail fwd: Y %.1f H1 %.1f H2 %.1f ln %.1f (logit -> tph8 %.1f) | tail bwd: sync %.1f ln-bwd %.1f conv1-dW %.1f sync %.1f dZ1 %.1f conv0 %.1f rest %.1f\n",
           tph[10] * 0.01, tph[11] * 0.01, tph[12] * 0.01, tph[13] * 0.01, tph[8] * 0.01, tph[14] * 0.01, tph[15] * 0.01, tph[16] * 0.01, tph[17] * 0.01, tph[18] * 0.01, tph[19] * 0.01, tph[9] * 0.01);
#endif
}

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
};
__global__ __launch_bounds__(512) void tail_slab_reduce_kernel(TailReduceArgs a) {
  __shared__ float4 part[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + lane, sp = blockIdx.y;
  int nt = a.count[a.count_idx];                       // tiles (or half tiles) planned by ragged.hip: every one of them wrote its slab
  if (nt > a.ntiles_cap) nt = a.ntiles_cap;
  const int lo = (int)((int64_t)nt * sp / kTailSplits), hi = (int)((int64_t)nt * (sp + 1) / kTailSplits);
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
  if (i < 8192) {
    // the two weight-gradient matrices arrive in the MFMA accumulator layout [wave][lane][register] (fused_fwd_kernel)
    const int e = i & 4095, wv = e >> 10, ln = (e >> 4) & 63, reg = e & 15;
    const int row = 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), col = 32 * (wv >> 1) + (ln & 31);
    a.dst[i >> 12][row * 64 + col] += s;
  } else {
    const int v = (i - kTailVec) >> 6, j = (i - kTailVec) & 63;
    a.dst[2 + v][j] += s;
  }
}

size_t fused_fold_floats() { return (size_t)3 * (MATCHA_N_HEAD * 64 * 64 + MATCHA_N_HEAD * 64); }

int launch_fold_ln(const matcha_tensors& p, float* ws, hipStream_t st) {
  FoldArgs a;
  const size_t wsz = (size_t)MATCHA_N_HEAD * 64 * 64, csz = (size_t)MATCHA_N_HEAD * 64;
  a.W[0] = p.w_q; a.W[1] = p.w_k; a.W[2] = p.w_v;
  a.g[0] = p.ln_q_g; a.g[1] = p.ln_k_g; a.g[2] = p.ln_v_g;
  a.b[0] = p.ln_q_b; a.b[1] = p.ln_k_b; a.b[2] = p.ln_v_b;
  for (int z = 0; z < 3; ++z) { a.Wp[z] = ws + z * wsz; a.c[z] = ws + 3 * wsz + z * csz; }
  hipLaunchKernelGGL(fold_ln_kernel, dim3(MATCHA_N_HEAD * 64, 3), dim3(64), 0, st, a);
  MATCHA_CHECK_LAUNCH("fold_ln_kernel");
  return MATCHA_OK;
}

size_t fused_tail_slab_floats(int64_t B, int L) { return (size_t)(ragged_tiles_cap(B, L) + 2) * kTailSlab; }
size_t fused_tail_partial_floats() { return (size_t)kTailSplits * kTailSlab; }
size_t fused_qkv_floats(int64_t B, int L) {
  const size_t tiles = (size_t)(ragged_tiles_cap(B, L) + 2) * MATCHA_N_HEAD * kImgRec;        // Q, K, V images per 64-row tile (or the merged r rows: kImgRecM)
  const size_t halves = (size_t)(ragged_halves_cap(B, L) + 2) * MATCHA_N_HEAD * kImgRecH;     // merged r rows per half tile
  return tiles > halves ? tiles : halves;
}

int launch_tail_reduce(const float* tslab, const Ragged& rg, int L, matcha_tensors& g_, hipStream_t st, bool halves, float* partial) {
  TailReduceArgs a;
  a.tslab = tslab; a.count = rg.count; a.L = L; a.ntiles_cap = halves ? rg.nhalves : rg.ntiles; a.count_idx = halves ? 3 : 2;
  a.partial = partial;
  float* dst[12] = {g_.pff1_w, g_.pff0_w, g_.pff_ln_g, g_.pff_ln_b, g_.ln1_g, g_.ln1_b, g_.ln2_g, g_.ln2_b, g_.cls_w, g_.pff1_b, g_.pff0_b, g_.cls_b};
  for (int i = 0; i < 12; ++i) a.dst[i] = dst[i];
  hipLaunchKernelGGL(tail_slab_reduce_kernel, dim3(kTailColBlocks, kTailSplits), dim3(512), 0, st, a);
  MATCHA_CHECK_LAUNCH("tail_slab_reduce_kernel");
  hipLaunchKernelGGL(tail_slab_finish_kernel, dim3((unsigned)cdiv(kTailSlab, 256)), dim3(256), 0, st, a);
  MATCHA_CHECK_LAUNCH("tail_slab_finish_kernel");
  return MATCHA_OK;
}

int launch_fused_fwd(const matcha_tensors& p, const float* folded, const float* X, const Ragged& rg, int64_t B, int L, const float* y, const float* w,
                     float* Y, float* H1, float* H2, float* logits, float* row_loss, const uint64_t* seed, float p_fc1, float p_pff,
                     hipStream_t st, float* ddyn0, float* dXs, float* tslab, float alpha, float* qkv) {
  FusedFwdArgs g;
  g.ddyn0 = (y && w) ? ddyn0 : nullptr; g.dXs = dXs; g.tslab = tslab; g.alpha_over_B = alpha / (float)B; g.qkv = qkv;
  const size_t wsz = (size_t)MATCHA_N_HEAD * 64 * 64, csz = (size_t)MATCHA_N_HEAD * 64;
  g.X = X; g.row_off = rg.row_off; g.tok_slot = rg.tok_slot; g.count = rg.count; g.tile_meta = rg.tile_meta; g.tok_pos = rg.tok_pos; g.B = B; g.L = L;
  g.wq = folded; g.wk = folded + wsz; g.wv = folded + 2 * wsz;
  g.cq = folded + 3 * wsz; g.ck = g.cq + csz; g.cv = g.cq + 2 * csz;
  g.fc1_w = p.fc1_w; g.fc1_b = p.fc1_b; g.p0w = p.pff0_w; g.p0b = p.pff0_b; g.p1w = p.pff1_w; g.p1b = p.pff1_b;
  g.hp = HeadParams{p.pff_ln_g, p.pff_ln_b, p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b, p.cls_w, p.cls_b};
  g.y = y; g.w = w; g.Y = Y; g.H1 = H1; g.H2 = H2; g.logits = logits; g.row_loss = (y && w) ? row_loss : nullptr;
  g.seed = seed; g.p_fc1 = p_fc1; g.p_pff = p_pff;
  g.dbg = options().fused_dbg;
  const int ntiles = rg.ntiles;
  size_t lds = ((size_t)4 * kTileF + 272 + 3 * 512 + 12 * 64) * sizeof(float);
  lds += (size_t)options().fwd_lds_pad;      // occupancy experiment: 1 workgroup per CU
  auto launch = [&](auto kfn) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kfn, dim3(ntiles), dim3(256), lds, st, g);
  };
  // algorithmic flops per token: 8 heads x 4 GEMMs (Q, K, V, fc1 block) + the two pff GEMMs, 2*64*64 each
  ProfScope ps(MATCHA_PROF_FUSED_FWD, (double)(B * L + 1) * (MATCHA_N_HEAD * 4.0 + 2.0) * 2.0 * 64.0 * 64.0, st);
  switch (L <= 2 ? 2 : (L <= 6 ? L : 8)) {
    case 2: launch(fused_fwd_kernel<2>); break;
    case 3: launch(fused_fwd_kernel<3>); break;
    case 4: launch(fused_fwd_kernel<4>); break;
    case 5: launch(fused_fwd_kernel<5>); break;
    case 6: launch(fused_fwd_kernel<6>); break;
    default: launch(fused_fwd_kernel<8>); break;
  }
  MATCHA_CHECK_LAUNCH("fused_fwd_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
