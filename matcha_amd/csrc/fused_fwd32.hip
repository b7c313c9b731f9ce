// Wave-independent fused forward for embed_dim 64, merged heads.  The mathematics of Modules.py:519-572,
// :353-376, :290-311 and main.py:56 with the per-head products merged (r = B_h x_hat + b_h, s_ij = r_i . x_hat_j, z_i = sum_j p_ij x_hat_j,
// dyn += M_h z_i), decomposed as
//
//   ONE WAVEFRONT = ONE WORKGROUP = one HALF TILE: <= 31 real tokens of whole hyperedges + the shared padding token as local row n
//   (ragged.hip: half_meta).  Nothing is shared between wavefronts, so there is no workgroup barrier anywhere (the four-wave tile kernel
//   of rounds 1-2 spent a third of its time in 59 barrier phases per tile).
//
//   Layout "FL": lane (r, h) owns token row r and the 32 features  f(e) = 32 wc + 8 g + 4 h + j,  e = 16 wc + 4 g + j  -- exactly what
//   the 32x32x2 MFMA leaves in a lane when a projection is computed as the TRANSPOSED product  D[feature][token] = W . x^T, and exactly
//   what the next product needs as its token-side operand.  r, z, dyn, Y, H1, H2 and their gradients therefore never leave the
//   registers; only the x_hat rows (keys = values of every head, read by the other tokens of a hyperedge) and the weight-gradient
//   operands pass through the wave's two private LDS tiles (2 x 8.5 KB: eight wavefronts per CU, two per SIMD, limited by the 256
//   registers).
//
//   Weights are STREAMED FROM L2 in fragment-major order (prep_heads_kernel rewrites them once per step: one coalesced 1 KB load per
//   wave and four MFMAs) through a rolling window of nine float4 per lane; tools/ubench/mfma_l2stream.hip measured 85 - 88 % of the
//   matrix pipe's rate for that pattern with two decoupled waves per SIMD, without any LDS staging.
//
//   Attention: two lanes per query token (32 features each), every token of the half tile at once; the dot products' cross-lane part
//   is ONE v_permlane32_swap instead of three DPP steps; cut into pieces that ride between the MFMAs of the next head's r product.
//
//   A training forward leaves, per (half tile, head), the r rows as the lane's accumulator registers + the probabilities [32][8]
//   (kImgRecH floats): fused_bwdh_kernel walks the same half tiles and reloads them thread for thread.
//
// fused_fwd32h_kernel (below) is the same forward for small batches: eight wavefronts per half tile, one per head.
#include "kernels.hpp"
#include "prep_heads.hpp"

#ifndef F32_ABL
#define F32_ABL 0                     // timing ablations (tools/debug/abl_fwd32.sh; results are wrong on purpose): 1 no attention pieces, 2 no record stores,
                                      // 4 no tail backward, 8 no tail, 32 no weight refills (stale window), 128 no operand splits
#endif

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kLdH = 68;                  // LDS row stride (floats)
constexpr int kHT = 32 * kLdH;            // one half tile [32 rows][68]
constexpr float kEps32 = 1e-5f;
constexpr int kTailVec32 = 2 * 4096;      // same slab format as fused_fwd.hip (tail_slab_reduce_kernel reads both)
constexpr int kTailSlab32 = 2 * 4096 + 10 * 64;

struct FL { f32x16 lo, hi; };             // 32 features of one token row in layout FL (lo: wc = 0, hi: wc = 1)

__device__ __forceinline__ float xhalf_sum(float v) {
  // lanes (r, 0) and (r, 1) hold the two halves of a row sum.  v_permlane32_swap a, b exchanges a[32:63] with b[0:31]: with a = b = v
  // it leaves a = {lo, lo}, b = {hi, hi}.  Inline asm: __builtin_amdgcn_permlane32_swap of this toolchain returns the FIRST result in
  // both elements (tools/debug/permlane_test.hip), i.e. only the lower half's broadcast.
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float fl_sum(const FL& a) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { s0 += a.lo[e]; s1 += a.hi[e]; }
  return s0 + s1;
}
__device__ __forceinline__ float fl_dot(const FL& a, const FL& b) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { s0 += a.lo[e] * b.lo[e]; s1 += a.hi[e] * b.hi[e]; }
  return s0 + s1;
}
// (mean, rstd) of a 64-feature row held as FL over two lanes
__device__ __forceinline__ void fl_stats(const FL& v, float& mean, float& rstd) {
  mean = xhalf_sum(fl_sum(v)) * (1.f / 64.f);
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { const float a = v.lo[e] - mean, b = v.hi[e] - mean; q0 += a * a; q1 += b * b; }
  rstd = __builtin_amdgcn_rsqf(xhalf_sum(q0 + q1) * (1.f / 64.f) + kEps32);
}
// row r of an LDS tile <-> FL registers (eight 16-byte accesses at columns 32 wc + 8 g + 4 h)
__device__ __forceinline__ void fl_store(float* __restrict__ rowp, const FL& v) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<f32x4*>(rowp + 8 * g) = (f32x4){v.lo[4 * g], v.lo[4 * g + 1], v.lo[4 * g + 2], v.lo[4 * g + 3]};
    *reinterpret_cast<f32x4*>(rowp + 32 + 8 * g) = (f32x4){v.hi[4 * g], v.hi[4 * g + 1], v.hi[4 * g + 2], v.hi[4 * g + 3]};
  }
}
__device__ __forceinline__ FL fl_load(const float* __restrict__ rowp) {
  FL v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + 8 * g), b = *reinterpret_cast<const f32x4*>(rowp + 32 + 8 * g);
    v.lo[4 * g] = a.x; v.lo[4 * g + 1] = a.y; v.lo[4 * g + 2] = a.z; v.lo[4 * g + 3] = a.w;
    v.hi[4 * g] = b.x; v.hi[4 * g + 1] = b.y; v.hi[4 * g + 2] = b.z; v.hi[4 * g + 3] = b.w;
  }
  return v;
}
// q . k over this lane's 32 features (packed pairs), k read from an LDS row
__device__ __forceinline__ float fl_dot_lds(const FL& q, const float* __restrict__ rowp) {
  f2 s = {0.f, 0.f}, t = {0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + 8 * g), b = *reinterpret_cast<const f32x4*>(rowp + 32 + 8 * g);
    s = __builtin_elementwise_fma((f2){q.lo[4 * g], q.lo[4 * g + 1]}, (f2){a.x, a.y}, s);
    t = __builtin_elementwise_fma((f2){q.lo[4 * g + 2], q.lo[4 * g + 3]}, (f2){a.z, a.w}, t);
    s = __builtin_elementwise_fma((f2){q.hi[4 * g], q.hi[4 * g + 1]}, (f2){b.x, b.y}, s);
    t = __builtin_elementwise_fma((f2){q.hi[4 * g + 2], q.hi[4 * g + 3]}, (f2){b.z, b.w}, t);
  }
  s += t;
  return s.x + s.y;
}
// o += w * row
__device__ __forceinline__ void fl_axpy_lds(FL& o, float w, const float* __restrict__ rowp) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + 8 * g), b = *reinterpret_cast<const f32x4*>(rowp + 32 + 8 * g);
    o.lo[4 * g] += w * a.x; o.lo[4 * g + 1] += w * a.y; o.lo[4 * g + 2] += w * a.z; o.lo[4 * g + 3] += w * a.w;
    o.hi[4 * g] += w * b.x; o.hi[4 * g + 1] += w * b.y; o.hi[4 * g + 2] += w * b.z; o.hi[4 * g + 3] += w * b.w;
  }
}
// v += vector (64 floats, global or LDS) at this lane's features
__device__ __forceinline__ void fl_add_vec(FL& v, const float* __restrict__ vec_h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(vec_h + 8 * g), b = *reinterpret_cast<const f32x4*>(vec_h + 32 + 8 * g);
    v.lo[4 * g] += a.x; v.lo[4 * g + 1] += a.y; v.lo[4 * g + 2] += a.z; v.lo[4 * g + 3] += a.w;
    v.hi[4 * g] += b.x; v.hi[4 * g + 1] += b.y; v.hi[4 * g + 2] += b.z; v.hi[4 * g + 3] += b.w;
  }
}
__device__ __forceinline__ FL fl_vec(const float* __restrict__ vec_h) {
  FL v;
#pragma unroll
  for (int e = 0; e < 16; ++e) { v.lo[e] = 0.f; v.hi[e] = 0.f; }
  fl_add_vec(v, vec_h);
  return v;
}
// rows of a [T, 64] global tensor <-> FL (16-byte accesses; a row's 256 bytes are covered by the two lanes in 8 instructions)
__device__ __forceinline__ void fl_store_global(float* __restrict__ rowp_h, const FL& v) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<f32x4*>(rowp_h + 8 * g) = (f32x4){v.lo[4 * g], v.lo[4 * g + 1], v.lo[4 * g + 2], v.lo[4 * g + 3]};
    *reinterpret_cast<f32x4*>(rowp_h + 32 + 8 * g) = (f32x4){v.hi[4 * g], v.hi[4 * g + 1], v.hi[4 * g + 2], v.hi[4 * g + 3]};
  }
}

// Reload of a row this lane parked earlier (fused_fwd32_tail.hpp): the offset passes through an empty asm, so the compiler cannot prove
// that the load reads what the park stored and forward the registers (which is the point of parking); the base keeps its address space.
// Per-lane values that are cheap functions of the lane id and of wave-uniform (scalar) values are REBUILT where they are used instead of
// living in a register across the kernel -- the token index, its row offset, the record offsets: at 256 registers the allocator spilled
// exactly those long-lived values.  The empty asm makes the lane id opaque at that point, so nothing is hoisted or kept.
__device__ __forceinline__ int f32_tok(int lane, int t0, int n, int tok_pad) {
  asm volatile("" : "+v"(lane));
  const int rr = lane & 31;
  return rr < n ? t0 + rr : tok_pad;                 // rows past the tokens compute on the shared padding token's copy (finite, unused)
}
// float offset of this lane's half row of its token in a [T, 64] tensor
__device__ __forceinline__ int64_t f32_row(int lane, int t0, int n, int tok_pad) {
  asm volatile("" : "+v"(lane));
  const int rr = lane & 31;
  return (int64_t)(rr < n ? t0 + rr : tok_pad) * 64 + 4 * (lane >> 5);
}
#define F32_TOK() f32_tok(lane, t0, n, tok_pad)
#define F32_ROW() f32_row(lane, t0, n, tok_pad)
__device__ __forceinline__ FL fl_unpark(const float* __restrict__ base, int64_t off) { return fl_load(base + off); }
__device__ __forceinline__ FL fl_unpark_lds(const float* __restrict__ base, int off) {
  asm volatile("" : "+v"(off));
  return fl_load(base + off);
}
// A wave-local ordering point for the tail: everything this wavefront has issued to the LDS is done (a wavefront's LDS operations execute
// in order anyway) and the compiler may not move memory accesses across it.  NOT a workgroup barrier: the tail is one wavefront's text.
#define F32_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#define MFMA32(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

// ---- the weight stream (round 5: fp32-accurate products on the bf16 matrix pipe) -------------------------------------------------------
// The f32 MFMA runs at the f32 VECTOR rate (64 FLOP / clk / SIMD, tools/ubench/mfma_rate.hip) and shares the vector ALUs with every other
// instruction of the wave (mfma_valu.hip: no co-issue), the bf16 MFMA at 16 x that rate on its own pipe.  Every operand is therefore split
// into THREE bf16 planes, v = h + m + l with h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) (round to nearest: |v - (h + m + l)| <=
// 2^-27 |v|), and a product is the six plane products whose magnitude exceeds 2^-26 of the result --
//     A B  ~=  Al Bh + Ah Bl + Am Bm + Am Bh + Ah Bm + Ah Bh          (dropped: Am Bl, Al Bm ~ 2^-27, Al Bl ~ 2^-36)
// -- each ONE v_mfma_f32_32x32x16_bf16 (K = 16, f32 accumulate; bf16 x bf16 products are exact in f32): 6 x 32 cycles instead of the
// 8 x 64 cycles of eight v_mfma_f32_32x32x2_f32 for the same 16 contraction indices, and the vector ALUs stay free for the attention
// arithmetic.  The result differs from an f32 fma chain by ~1e-7 relative, the size of that chain's own rounding (tests compare with the
// oracle at the same tolerances as before; tests/test_cpu_bf16x3.py restates the arithmetic in numpy).
//
// The WEIGHT side is split once per step by prep_heads_kernel; one 64 x 64 matrix = 24 fragments of one u32x4 (eight bf16) per lane, in
// consumption order [chunk c = 0..3][output block wc = 0, 1][plane h, m, l]: lane (r, hf), element j holds
//     W[32 wc + r][16 c + 8 (j >> 2) + 4 hf + (j & 3)]
// -- the contraction slot k = 8 hf + j of the MFMA paired with the feature that layout FL keeps in register 8 c + j of lane half hf, so
// the TOKEN side (a register-resident FL row) needs no shuffle: its chunk c is split in registers (36 vector instructions per eight
// values) right before the two steps that use it; x_hat, the operand of all eight r products, is split once per half tile.  Biases are
// f32 vectors added to the accumulator (kBias*: b_0..b_7, the merged fc1 bias, conv0, conv1), not part of the stream.
// The window W_[0..5] holds the six fragments of the current chunk; every consumed step refills its three slots with the fragments six
// ahead (`wp`, wave-uniform, points at the PREFETCH position), pinned there by a scheduling barrier.
#define MFMA_BF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)), __builtin_bit_cast(bf16x8, (B)), (C), 0, 0, 0)
#ifndef F32_WIN
#define F32_WIN 6                     // fragments in flight ahead of the product (divides 24)
#endif

struct B3 { u32x4 h, m, l; };                   // eight f32 values as three bf16 planes
// register 8 C + J of an FL row (C = 0..3 compile-time): features 16 C + 8 (J >> 2) + 4 hf + (J & 3)
#define FL_CHUNK(V, C, J) ((C) < 2 ? (V).lo[8 * ((C) & 1) + (J)] : (V).hi[8 * ((C) & 1) + (J)])
#define FL_SPLIT(OUT, V, C)                                                                              \
  do {                                                                                                   \
    _Pragma("unroll") for (int q__ = 0; q__ < 4; ++q__) {                                                \
      if (F32_ABL & 128) { (OUT).h[q__] = __builtin_bit_cast(uint32_t, FL_CHUNK(V, C, 2 * q__)); (OUT).m[q__] = (OUT).h[q__]; (OUT).l[q__] = (OUT).h[q__]; continue; } \
      const P3 p__ = split2(FL_CHUNK(V, C, 2 * q__), FL_CHUNK(V, C, 2 * q__ + 1));                       \
      (OUT).h[q__] = p__.h; (OUT).m[q__] = p__.m; (OUT).l[q__] = p__.l;                                  \
    }                                                                                                    \
  } while (0)
// step S = 2 c + wc of a product: the six plane products of (chunk c, output block wc), smallest terms first
#define WB_MMA(ACC, XB, S)                                                                               \
  do {                                                                                                   \
    const u32x4 ah__ = W_[(3 * (S)) % F32_WIN], am__ = W_[(3 * (S) + 1) % F32_WIN], al__ = W_[(3 * (S) + 2) % F32_WIN]; \
    if constexpr (((S) & 1) == 0) {                                                                      \
      ACC.lo = MFMA_BF(al__, (XB).h, ACC.lo); ACC.lo = MFMA_BF(ah__, (XB).l, ACC.lo); ACC.lo = MFMA_BF(am__, (XB).m, ACC.lo); \
      ACC.lo = MFMA_BF(am__, (XB).h, ACC.lo); ACC.lo = MFMA_BF(ah__, (XB).m, ACC.lo); ACC.lo = MFMA_BF(ah__, (XB).h, ACC.lo); \
    } else {                                                                                             \
      ACC.hi = MFMA_BF(al__, (XB).h, ACC.hi); ACC.hi = MFMA_BF(ah__, (XB).l, ACC.hi); ACC.hi = MFMA_BF(am__, (XB).m, ACC.hi); \
      ACC.hi = MFMA_BF(am__, (XB).h, ACC.hi); ACC.hi = MFMA_BF(ah__, (XB).m, ACC.hi); ACC.hi = MFMA_BF(ah__, (XB).h, ACC.hi); \
    }                                                                                                    \
  } while (0)
#define WB_REFILL(S, PF)                                                                                 \
  do {                                                                                                   \
    if ((PF) && !(F32_ABL & 32)) {                                                                       \
      W_[(3 * (S)) % F32_WIN] = wp[lane]; W_[(3 * (S) + 1) % F32_WIN] = (wp + 64)[lane]; W_[(3 * (S) + 2) % F32_WIN] = (wp + 128)[lane]; \
    }                                                                                                    \
    wp += 192;                                                                                           \
    asm volatile("" ::: "memory");         /* instruction selection clusters every load of the product at its top otherwise */ \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
// a whole product ACC += W . B^T with nothing interleaved, B an FL row in f32 (split chunk by chunk on the way); PF_LAST = false stops
// refilling in the last chunk (the caller re-primes the window at another matrix: W32_PRIME_AT)
#define WB_CHUNK(ACC, B, C, PF)                                                                          \
  do {                                                                                                   \
    B3 xb__;                                                                                             \
    FL_SPLIT(xb__, B, C);                                                                                \
    WB_MMA(ACC, xb__, 2 * (C)); WB_REFILL(2 * (C), PF); WB_MMA(ACC, xb__, 2 * (C) + 1); WB_REFILL(2 * (C) + 1, PF); \
  } while (0)
#define W32_CHAIN(ACC, B, PF_LAST)                                                                       \
  do { WB_CHUNK(ACC, B, 0, true); WB_CHUNK(ACC, B, 1, true); WB_CHUNK(ACC, B, 2, true); WB_CHUNK(ACC, B, 3, PF_LAST); } while (0)
// the same with B already split (x_hat)
#define W32_CHAIN_XS(ACC, XS, PF_LAST)                                                                   \
  do {                                                                                                   \
    WB_MMA(ACC, XS[0], 0); WB_REFILL(0, true); WB_MMA(ACC, XS[0], 1); WB_REFILL(1, true);                \
    WB_MMA(ACC, XS[1], 2); WB_REFILL(2, true); WB_MMA(ACC, XS[1], 3); WB_REFILL(3, true);                \
    WB_MMA(ACC, XS[2], 4); WB_REFILL(4, true); WB_MMA(ACC, XS[2], 5); WB_REFILL(5, true);                \
    WB_MMA(ACC, XS[3], 6); WB_REFILL(6, PF_LAST); WB_MMA(ACC, XS[3], 7); WB_REFILL(7, PF_LAST);          \
  } while (0)
// window <- the first six fragments of matrix MAT of the stream; wp = the prefetch position behind them
#define W32_PRIME_AT(MAT)                                                                                \
  do {                                                                                                   \
    wp = g.wfrag + (MAT) * kFragU4;                                                                      \
    _Pragma("unroll") for (int i__ = 0; i__ < F32_WIN; ++i__) W_[i__] = (wp + i__ * 64)[lane];           \
    wp += F32_WIN * 64;                                                                                  \
  } while (0)

__device__ __forceinline__ FL fl_zero() {
  FL v;
#pragma unroll
  for (int e = 0; e < 16; ++e) { v.lo[e] = 0.f; v.hi[e] = 0.f; }
  return v;
}

}  // namespace

// ---- merged per-head matrices (round 3) -----------------------------------------------------------------------------------
// With d_k = d_v = d_model (MATCHA's configuration, main.py:615-623) nothing non-linear sits between a projection and the product
// that consumes it, so per head the four 64 x 64 products of the reference (Modules.py:527-529, :572) collapse into two:
//   scores   q_i . k_j = x_j^T (W'k^T W'q x_i + W'k^T cq) + terms that do not depend on j (they cancel in the softmax, whose masked
//            diagonal is REPLACED, Modules.py:443-445)           ->  r_i = B_h x_i + b_h,   s_ij = r_i . x_j
//   output   sum_j p_ij v_j = W'v (sum_j p_ij x_j) + cv  (the probabilities, padding slots included, sum to 1)
//            dyn += Wfc1_h O_i                                   ->  z_i = sum_j p_ij x_j,  dyn += M_h z_i,   M_h = Wfc1_h W'v_h
// (x = the LayerNorm-normalised row, the same for every head: it is the key AND the value of every head, written to LDS once per tile.)
// The constants Wfc1_h cv_h join the fc1 bias.  prep_heads_kernel builds B_h, b_h, M_h and that bias once per step.
// ---- the per-step weight forms (prep_heads.hpp): stand-alone launch for the paths whose front end is not front_fwd_kernel ----------------
__global__ __launch_bounds__(256) void prep_heads_kernel(PrepArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[kPrepLdsFloats];
  prep_heads_role(a, blockIdx.x, blockIdx.y, blockIdx.z, sm);
}

struct Fwd32Args {
  const float* X;
  const int32_t* row_off; const int32_t* tok_slot; const int32_t* count; const int32_t* half_meta; const int32_t* tok_pos;
  int L;
  const u32x4* wfrag;              // the bf16 x 3 fragment stream + the f32 bias table behind it (prep_heads_kernel)
  HeadParams hp;
  const float* y; const float* w;
  float* Y; float* H1; float* H2;
  float* logits; float* row_loss;
  const uint64_t* seed;
  float p_fc1, p_pff;
  float* ddyn0; float* dXs; float* tslab; float alpha_over_B;
  float* qkv;         // training: the record per (half tile, head) -- this wavefront's own r rows + probabilities, kImgRecH floats (fused_bwdh_kernel)
  float* tail_dh2;    // single-wave kernel, training: [T][64] dH2 rows -- the tail's backward stops behind its LayerNorms, tail_bwd64_kernel does the convolutions
};

// Merged per-head matrices (two products per head: r = B_h x + b_h, dyn += M_h z).
template <int ML>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fused_fwd32_kernel(Fwd32Args g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef FF_TIMING
  long long tph[16] = {0};
  long long tlast = wall_clock64();
#define FF_T(i) do { const long long now__ = wall_clock64(); tph[i] += now__ - tlast; tlast = now__; } while (0)
#else
#define FF_T(i) do { } while (0)
#endif
  float* TK = lds;                       // the x_hat rows (keys = values of every head); tail: product tiles, dH2, Y
  float* TV = lds + kHT;                 // tail: the parameter vectors until the weight-gradient GEMMs, then H1, dZ1
  float* outs = lds + 2 * kHT;           // [32] per-token classifier outputs
  float* douts = outs + 32;              // [32] their gradients
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;

  const int4 meta = reinterpret_cast<const int4*>(g.half_meta)[blockIdx.x];
  const int t0 = meta.x, n = meta.y, b0 = meta.z, n_h = meta.w;       // n <= 31 tokens, local row n = the shared padding token
  if (n_h <= 0) return;
  const int tok_pad = g.count[1];
  const float inv_temp = 0.125f;
  const bool real = r < n;

  // ---- weight stream: prime the window with the first chunk of R_0; the f32 bias table -> TV (free until the tail) ----
  const u32x4* wp;
  u32x4 W_[F32_WIN];
  W32_PRIME_AT(0);
  {
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(g.wfrag + (kNMat + 1) * kFragU4);
    for (int i4 = lane; i4 < kNBias * 16; i4 += 64) reinterpret_cast<f32x4*>(TV)[i4] = bsrc[i4];
  }
#define F32_BIAS(ROW) fl_vec(TV + (ROW) * 64 + 4 * h)

  // ---- x_hat in layout FL straight from global memory ----
  FL xh = fl_load(g.X + F32_ROW());
  int pos = 0, k = 0;
  if (real) {
    const int tp = g.tok_pos[F32_TOK()];
    pos = tp & 255; k = tp >> 8;
  }
  const int li0 = r - pos;
  float rx;
  {
    float mean;
    fl_stats(xh, mean, rx);
#pragma unroll
    for (int e = 0; e < 16; ++e) { xh.lo[e] = (xh.lo[e] - mean) * rx; xh.hi[e] = (xh.hi[e] - mean) * rx; }
  }
  constexpr int kRec = kImgRecH;
  // where this token's rows go in the backward kernel's record: float4 index (wc * 4 + g) * 64 + 32 h + r of this wavefront's own
  // record, probabilities [32][8] behind the rows
  // (wave-uniform base + 32-bit lane offsets; two per-lane pointers held across the head loop were spill candidates)
  float* const img_base = g.qkv ? g.qkv + (int64_t)blockIdx.x * MATCHA_N_HEAD * kImgRecH : nullptr;
  const bool img_on = g.qkv != nullptr && real;
#define F32_IMG_STORE(ACC, HD, M)                                                                        \
  do {                                                                                                   \
    if (img_on && !(F32_ABL & 2)) {                                                                      \
      int l__ = lane;                                                                                    \
      asm volatile("" : "+v"(l__));                                                                      \
      f32x4* d__ = reinterpret_cast<f32x4*>(img_base + (HD) * kRec + (M) * 4096 + l__ * 4);             \
      _Pragma("unroll") for (int g__ = 0; g__ < 4; ++g__) {                                              \
        if (F32_ABL & 16) {                                                                              \
          d__[g__ * 64] = (f32x4){ACC.lo[4 * g__], ACC.lo[4 * g__ + 1], ACC.lo[4 * g__ + 2], ACC.lo[4 * g__ + 3]};   \
          d__[(4 + g__) * 64] = (f32x4){ACC.hi[4 * g__], ACC.hi[4 * g__ + 1], ACC.hi[4 * g__ + 2], ACC.hi[4 * g__ + 3]}; \
        } else {                                                                                         \
        __builtin_nontemporal_store((f32x4){ACC.lo[4 * g__], ACC.lo[4 * g__ + 1], ACC.lo[4 * g__ + 2], ACC.lo[4 * g__ + 3]}, d__ + g__ * 64); \
        __builtin_nontemporal_store((f32x4){ACC.hi[4 * g__], ACC.hi[4 * g__ + 1], ACC.hi[4 * g__ + 2], ACC.hi[4 * g__ + 3]}, d__ + (4 + g__) * 64); \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)

  const int n_pad = g.L - k;
  const float padf = (float)n_pad;
  const bool hpad = n_pad > 0;
  int ro[ML + 1];                                     // LDS offset of key / value row j of this token's hyperedge (slots j >= k: clamped); [ML] = the padding token's row
#pragma unroll
  for (int j = 0; j < ML; ++j) ro[j] = (li0 + (j < k ? j : 0)) * kLdH + 4 * h;
  ro[ML] = n * kLdH + 4 * h;
  float* krow = TK + r * kLdH + 4 * h;
  float* vrow = TV + r * kLdH + 4 * h;

  // Attention of one head is cut into PIECES of one half key / value row (4 LDS reads + 8 packed FMAs) that ride between the steps of a
  // product that does not depend on them: the scores of head h between the MFMAs of V_h, P V of head h between those of K_{h+1}.
  // Instruction selection places pure arithmetic wherever it likes inside the basic block -- it sank every piece's FMAs to the end of
  // the stage and spilled the rows they wait for; an empty asm that takes the piece's results as in/out operands pins them to the step.
#define F32_PIN1(V) asm volatile("" : "+v"(V))
  float p[ML + 1];                                    // scores, then probabilities x 1 / denominator ([ML]: all padding slots together)
  float sc_part = 0.f;
#define F32_SC_PIECE(S)                                                                                  \
  do {                                                                                                   \
    constexpr int j__ = (S) >> 1, hf__ = (S) & 1;                                                        \
    if constexpr (j__ <= ML) {                                                                           \
      const float* rp__ = TK + ro[j__] + 32 * hf__;                                                      \
      f2 s__ = {0.f, 0.f}, t__ = {0.f, 0.f};                                                             \
      _Pragma("unroll") for (int g__ = 0; g__ < 4; ++g__) {                                              \
        const f32x4 a__ = *reinterpret_cast<const f32x4*>(rp__ + 8 * g__);                               \
        const f32x16& qq__ = hf__ ? q.hi : q.lo;                                                         \
        s__ = __builtin_elementwise_fma((f2){qq__[4 * g__], qq__[4 * g__ + 1]}, (f2){a__.x, a__.y}, s__); \
        t__ = __builtin_elementwise_fma((f2){qq__[4 * g__ + 2], qq__[4 * g__ + 3]}, (f2){a__.z, a__.w}, t__); \
      }                                                                                                  \
      s__ += t__;                                                                                        \
      if constexpr (hf__ == 0) { sc_part = s__.x + s__.y; F32_PIN1(sc_part); }                            \
      else {                                                                                             \
        float a__ = xhalf_sum(sc_part + (s__.x + s__.y)) * inv_temp;                                     \
        if constexpr (j__ < ML) a__ = (j__ == pos) ? -1e32f : a__;     /* masked diagonal (Modules.py:443-445) */ \
        p[j__] = a__;                                                                                    \
        F32_PIN1(p[j__]);                                                                                \
      }                                                                                                  \
    }                                                                                                    \
    if constexpr ((S) == 2 * (ML + 1)) {                                                                 \
      float mx__ = -3.4e38f;                                                                             \
      _Pragma("unroll") for (int i__ = 0; i__ < ML; ++i__) mx__ = (i__ < k) ? fmaxf(mx__, p[i__]) : mx__; \
      mx__ = hpad ? fmaxf(mx__, p[ML]) : mx__;                                                           \
      float den__ = 0.f;                                                                                 \
      _Pragma("unroll") for (int i__ = 0; i__ < ML; ++i__) { p[i__] = (i__ < k) ? __expf(p[i__] - mx__) : 0.f; den__ += p[i__]; } \
      p[ML] = hpad ? __expf(p[ML] - mx__) : 0.f;                                                         \
      den__ += padf * p[ML];                                                                             \
      const float inv__ = __builtin_amdgcn_rcpf(den__);                                                  \
      _Pragma("unroll") for (int i__ = 0; i__ <= ML; ++i__) { p[i__] *= inv__; F32_PIN1(p[i__]); }       \
    }                                                                                                    \
    if constexpr ((S) == 2 * (ML + 1) + 1) {                                                             \
      /* training: row i of P for the backward pass -- slots 0..k-1 the real keys, slot 7 the per-slot padding probability */ \
      if (img_on && h == 0) {                                                                            \
        float wv__[8];                                                                                   \
        _Pragma("unroll") for (int i__ = 0; i__ < 8; ++i__) wv__[i__] = i__ < ML ? p[i__ < ML ? i__ : 0] : 0.f; \
        if (hpad) wv__[7] = p[ML];                                                                       \
        int l__ = lane;                                                                                  \
        asm volatile("" : "+v"(l__));                                                                    \
        f32x4* dst__ = reinterpret_cast<f32x4*>(img_base + hd * kRec + 2048 + (l__ & 31) * 8);          \
        __builtin_nontemporal_store((f32x4){wv__[0], wv__[1], wv__[2], wv__[3]}, dst__);                 \
        __builtin_nontemporal_store((f32x4){wv__[4], wv__[5], wv__[6], wv__[7]}, dst__ + 1);             \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
#define F32_STAGE8(MAC) do { MAC(0); MAC(1); MAC(2); MAC(3); MAC(4); MAC(5); MAC(6); MAC(7); } while (0)

  FF_T(0);
  fl_store(krow, xh);
  // x_hat as three bf16 planes: the token-side operand of all eight r products (x_hat in f32 is dead until the tail, which reads its row back)
  B3 xs[4];
  FL_SPLIT(xs[0], xh, 0); FL_SPLIT(xs[1], xh, 1); FL_SPLIT(xs[2], xh, 2); FL_SPLIT(xs[3], xh, 3);
  F32_WAVE_SYNC();                                    // x_hat rows and the bias table visible (one wavefront = the whole workgroup)
  FL dyn = F32_BIAS(kBiasDyn);                        // the merged fc1 bias enters once
  {
    // ======================= merged heads: two products per head, keys = values = the x_hat rows (written to TK once) =======================
    // pieces of head hd (five per step of the NEXT head's r product): score half-dots, softmax, probabilities out, z half-rows
    FL q = F32_BIAS(kBiasR), o;                       // q: the r rows of the current head; o: z = P x_hat
    W32_CHAIN_XS(q, xs, true);                        // r_0 = B_0 x_hat + b_0
    F32_IMG_STORE(q, 0, 0);
    FF_T(1);
#define F32_Z_PIECE(S)                                                                                   \
  do {                                                                                                   \
    constexpr int j__ = (S) >> 1, hf__ = (S) & 1;                                                        \
    if constexpr (j__ <= ML) {                                                                           \
      const float* rp__ = TK + ro[j__] + 32 * hf__;                                                      \
      const float w__ = j__ < ML ? p[j__] : padf * p[ML];                                                \
      const f2 w2__ = {w__, w__};                                                                        \
      f32x16& oo__ = hf__ ? o.hi : o.lo;                                                                 \
      _Pragma("unroll") for (int g__ = 0; g__ < 4; ++g__) {                                              \
        const f32x4 a__ = *reinterpret_cast<const f32x4*>(rp__ + 8 * g__);                               \
        const f2 u0__ = __builtin_elementwise_fma(w2__, (f2){a__.x, a__.y}, (f2){oo__[4 * g__], oo__[4 * g__ + 1]});     \
        const f2 u1__ = __builtin_elementwise_fma(w2__, (f2){a__.z, a__.w}, (f2){oo__[4 * g__ + 2], oo__[4 * g__ + 3]}); \
        oo__[4 * g__] = u0__.x; oo__[4 * g__ + 1] = u0__.y; oo__[4 * g__ + 2] = u1__.x; oo__[4 * g__ + 3] = u1__.y;      \
      }                                                                                                  \
      asm volatile("" : "+v"(oo__));                                                                     \
    }                                                                                                    \
  } while (0)
    // piece index Q: [0, 2 (ML + 1) + 2): the score pieces of F32_SC_PIECE (dots, softmax, probabilities out), then the z half-rows
#define F32_MG_PIECE(Q)                                                                                  \
  do {                                                                                                   \
    if constexpr ((F32_ABL & 1) != 0) { } else                                                           \
    if constexpr ((Q) < 2 * (ML + 1) + 2) F32_SC_PIECE(Q); else F32_Z_PIECE((Q) - (2 * (ML + 1) + 2));   \
  } while (0)
    // ML = 8: 4 ML + 6 = 38 pieces over the eight steps of a product, five per step
#define F32_MG_ONLY(S) do { F32_MG_PIECE(5 * (S)); F32_MG_PIECE(5 * (S) + 1); F32_MG_PIECE(5 * (S) + 2); F32_MG_PIECE(5 * (S) + 3); F32_MG_PIECE(5 * (S) + 4); } while (0)
#define F32_MG_STEP(S) do { WB_MMA(acc, xs[(S) >> 1], S); F32_MG_ONLY(S); WB_REFILL(S, true); } while (0)
    int hd = 0;
    for (; hd + 1 < MATCHA_N_HEAD; ++hd) {
      o = fl_zero();
      {
        FL acc = F32_BIAS(kBiasR + hd + 1);           // r_{hd+1} = B_{hd+1} x_hat + b_{hd+1}
        F32_STAGE8(F32_MG_STEP);
        q = acc;
        F32_IMG_STORE(q, hd + 1, 0);
      }
      FF_T(2);
      W32_CHAIN(dyn, o, true);                        // dyn += M_hd z
      FF_T(5);
    }
    o = fl_zero();
    F32_STAGE8(F32_MG_ONLY);
    FF_T(2);
    W32_CHAIN(dyn, o, true);
    FF_T(5);
  }
  // =========================== tail: pff_n1, LayerNorms, classifier (all in registers) ===========================
  if (F32_ABL & 8) { if (dyn.lo[0] == 12345.f) g.logits[0] = dyn.hi[3]; return; }
  // Y and H1 are parked in the workspace's Y / H1 rows (every lane stores: the rows past the tokens are copies of the padding token's row,
  // identical in every lane and in every half tile) -- in a saving forward they are what the layer-wise backward reads anyway
  // x_hat in f32 is read back from this lane's own row of TK where the tail first needs it (the head loop works on its bf16 planes)
#define F32_XH_RELOAD() do { xh = fl_load(krow); } while (0)
#define F32_TAIL_SYNC() F32_WAVE_SYNC()
#define F32_PARK_Y(V) do { if (g.Y) fl_store_global(g.Y + F32_ROW(), V); } while (0)
#define F32_PARK_H1(V) do { if (g.H1) fl_store_global(g.H1 + F32_ROW(), V); } while (0)
#define F32_UNPARK_Y() fl_unpark(g.Y, F32_ROW())
#define F32_UNPARK_H1() fl_unpark(g.H1, F32_ROW())
#define F32_PARK_HH(V) do { if (g.H2) fl_store_global(g.H2 + F32_ROW(), V); } while (0)
#define F32_UNPARK_HH() fl_unpark(g.H2, F32_ROW())
#include "fused_fwd32_tail.hpp"
#undef F32_TAIL_SYNC
#undef F32_PARK_Y
#undef F32_PARK_H1
#undef F32_UNPARK_Y
#undef F32_UNPARK_H1
#undef F32_PARK_HH
#undef F32_UNPARK_HH
#ifdef FF_TIMING
  if (blockIdx.x == 2000 && lane == 0)
    printf("fused_fwd32 wave 2000 us: setup %.1f prologue K0 Q0 %.1f | 8 heads: V+scores %.1f K'+PV %.1f Q' %.1f fc1 %.1f | pff fwd %.1f ln+logit %.1f | ln-bwd+colsums %.1f dW1 %.1f dZ1 %.1f dW0 %.1f ddyn %.1f\n",
           tph[0] * 0.01, tph[1] * 0.01, tph[2] * 0.01, tph[3] * 0.01, tph[4] * 0.01, tph[5] * 0.01, tph[7] * 0.01, tph[8] * 0.01, tph[9] * 0.01,
           tph[10] * 0.01, tph[11] * 0.01, tph[12] * 0.01, tph[13] * 0.01);
#endif
}

// ---- the same forward for SMALL batches: eight wavefronts per half tile, one per head ---------------------------------------------------
// fused_fwd32_kernel gives a half tile to ONE wavefront, which walks the eight heads in turn: 78 us of latency whatever the batch.  At the
// reference's own batch (96 + 288 rows, main.py:527-528: ~45 half tiles on 256 CUs) that chain IS the kernel's duration.  Here a workgroup
// of eight wavefronts takes the half tile: every wavefront normalises the rows (redundantly: 32 in-lane adds), wavefront hd computes
// r = B_hd x_hat + b_hd, the attention of head hd on the shared x_hat tile and its dyn contribution M_hd z, the eight partial dyn tiles
// are summed in head order by wavefront 0, which runs the tail alone (the same text: fused_fwd32_tail.hpp).  Merged heads only; the saved
// records, the slab of parameter-gradient partials and every output are exactly the single-wave kernel's, so the backward kernels do not
// know which forward ran.  The heads' partial products are summed as eight separately rounded tiles instead of one MFMA accumulation
// chain: logits differ from the single-wave kernel by ~1e-7 relative (tests/test_hip_properties.py).
template <int ML>
__global__ __launch_bounds__(512) void fused_fwd32h_kernel(Fwd32Args g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#undef FF_T
#define FF_T(i) do { } while (0)
  float* TK = lds;                       // the x_hat rows (keys = values of every head); tail: product tiles
  float* TV = lds + kHT;                 // tail: parameter vectors, H1, dZ1
  float* outs = lds + 2 * kHT;
  float* douts = outs + 32;
  float* Pd = lds + 2 * kHT + 64;        // [8][32][kLdH] the heads' dyn contributions
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int hd = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // this wavefront's head (wave-uniform: a scalar register)

  const int4 meta = reinterpret_cast<const int4*>(g.half_meta)[blockIdx.x];
  const int t0 = meta.x, n = meta.y, b0 = meta.z, n_h = meta.w;
  if (n_h <= 0) return;
  const int tok_pad = g.count[1];
  const float inv_temp = 0.125f;
  const bool real = r < n;

  // weight stream of THIS head: R_hd sits at matrix 0 (hd = 0) or 2 hd - 1, M_hd at 2 hd + 2 (hd < 7) or 15 (prep_heads_kernel's stream order)
  const u32x4* wp;
  u32x4 W_[F32_WIN];
  W32_PRIME_AT(hd == 0 ? 0 : 2 * hd - 1);
#undef F32_BIAS
#define F32_BIAS(ROW) fl_vec(reinterpret_cast<const float*>(g.wfrag + (kNMat + 1) * kFragU4) + (ROW) * 64 + 4 * h)      /* from L2: two per wavefront */

  FL xh = fl_load(g.X + F32_ROW());
  int pos = 0, k = 0;
  if (real) {
    const int tp = g.tok_pos[F32_TOK()];
    pos = tp & 255; k = tp >> 8;
  }
  const int li0 = r - pos;
  float rx;
  {
    float mean;
    fl_stats(xh, mean, rx);
#pragma unroll
    for (int e = 0; e < 16; ++e) { xh.lo[e] = (xh.lo[e] - mean) * rx; xh.hi[e] = (xh.hi[e] - mean) * rx; }
  }
  const int kRec = kImgRecH;
  float* const img_base = g.qkv ? g.qkv + (int64_t)blockIdx.x * MATCHA_N_HEAD * kImgRecH : nullptr;     // half-tile records: as in fused_fwd32_kernel
  const bool img_on = g.qkv != nullptr && real;
  const int n_pad = g.L - k;
  const float padf = (float)n_pad;
  const bool hpad = n_pad > 0;
  int ro[ML + 1];
#pragma unroll
  for (int j = 0; j < ML; ++j) ro[j] = (li0 + (j < k ? j : 0)) * kLdH + 4 * h;
  ro[ML] = n * kLdH + 4 * h;
  float* krow = TK + r * kLdH + 4 * h;
  float p[ML + 1];
  float sc_part = 0.f;

  if (hd == 0) fl_store(krow, xh);
  FL q = F32_BIAS(kBiasR + hd), o = fl_zero();
  W32_CHAIN(q, xh, false);                            // r_hd = B_hd x_hat + b_hd
  F32_IMG_STORE(q, hd, 0);
  W32_PRIME_AT(hd < 7 ? 2 * hd + 2 : 15);             // M_hd: in flight during the attention
  __syncthreads();                                    // x_hat rows visible
  F32_STAGE8(F32_MG_ONLY);                            // scores, softmax, probabilities out, z = P x_hat
  {
    FL dynp = fl_zero();
    if (hd == 0) dynp = F32_BIAS(kBiasDyn);           // the merged fc1 bias enters once, with head 0
    W32_CHAIN(dynp, o, false);                        // M_hd z
    fl_store(Pd + hd * kHT + r * kLdH + 4 * h, dynp);
  }
  __syncthreads();                                    // the LAST workgroup barrier of this kernel
  if (hd != 0) return;                                // wavefront 0 alone from here: the tail's ordering points are wave-local (F32_WAVE_SYNC)
  FL dyn = fl_load(Pd + r * kLdH + 4 * h);
#pragma unroll 1
  for (int j = 1; j < MATCHA_N_HEAD; ++j) {
    const FL t = fl_load(Pd + j * kHT + r * kLdH + 4 * h);
#pragma unroll
    for (int e = 0; e < 16; ++e) { dyn.lo[e] += t.lo[e]; dyn.hi[e] += t.hi[e]; }
  }
  W32_PRIME_AT(16);                                   // conv0, conv1, conv1^T, conv0^T follow the heads in the stream
  // one wavefront is left, so the tail needs no workgroup barrier (round 4 called __syncthreads() here after seven of the eight wavefronts
  // had returned: it worked -- terminated wavefronts no longer count -- but it is outside HIP's barrier contract).  Y and H1 are parked in
  // the first two of the heads' partial-product tiles, which are dead once `dyn` has been summed; a saving forward also writes them out.
#define F32_TAIL_SYNC() F32_WAVE_SYNC()
#define F32_PARK_Y(V) do { fl_store(Pd + r * kLdH + 4 * h, V); if (g.Y && !g.ddyn0 && r <= n) fl_store_global(g.Y + F32_ROW(), V); } while (0)
#define F32_PARK_H1(V) do { fl_store(Pd + kHT + r * kLdH + 4 * h, V); if (g.H1 && !g.ddyn0 && r <= n) fl_store_global(g.H1 + F32_ROW(), V); } while (0)
#define F32_UNPARK_Y() fl_unpark_lds(Pd, r * kLdH + 4 * h)
#define F32_UNPARK_H1() fl_unpark_lds(Pd + kHT, r * kLdH + 4 * h)
#define F32_PARK_HH(V) fl_store(Pd + 2 * kHT + r * kLdH + 4 * h, V)
#define F32_UNPARK_HH() fl_unpark_lds(Pd + 2 * kHT, r * kLdH + 4 * h)
#include "fused_fwd32_tail.hpp"
#undef F32_TAIL_SYNC
#undef F32_PARK_Y
#undef F32_PARK_H1
#undef F32_UNPARK_Y
#undef F32_UNPARK_H1
#undef F32_PARK_HH
#undef F32_UNPARK_HH
#ifdef F32H_REPRO
  // tools/debug/fwd32h_repro.sh: round 4's failing variant -- an UNREACHABLE block with a fence, an atomic, a wave shuffle
  // and a loop appended to the kernel.  It never runs (L <= MATCHA_MAX_L); what it changes is the register allocation of everything above.
  if (g.L == 0x40000000) {
    __threadfence();
    unsigned int* ctr = reinterpret_cast<unsigned int*>(g.tslab);
    unsigned int t = 0;
    if (lane == 0) t = atomicAdd(ctr, 1u);
    t = __shfl(t, 0, 64);
    if (t == gridDim.x - 1) {
      float s = 0.f;
      for (int i = lane; i < g.L * 977; i += 64) s += g.row_loss[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) g.logits[0] = s;
    }
  }
#endif
}

static int fwd32h_max_halves() { return 2 * device_cu_count(); }

bool fused_small_batch(const Ragged& rg) { return rg.nhalves <= fwd32h_max_halves() && !options().disable_small_batch; }

size_t fused_frag_floats() { return (size_t)(kNMat + 1) * kFragU4 * 4 + (size_t)kNBias * 64; }
size_t fused_tail_slab32_floats(int64_t B, int L) { return (size_t)(ragged_halves_cap(B, L) + 2) * kTailSlab32; }

size_t fused_merged_floats() { return (size_t)2 * MATCHA_N_HEAD * 4096 + MATCHA_N_HEAD * 64 + 64; }
MergedView merged_view(const float* m) {
  return MergedView{m, m + (size_t)MATCHA_N_HEAD * 4096, m + (size_t)2 * MATCHA_N_HEAD * 4096, m + (size_t)2 * MATCHA_N_HEAD * 4096 + MATCHA_N_HEAD * 64};
}
// the per-step weight forms (prep_heads_kernel): `folded` = fused_fold_floats() floats, `merged` = fused_merged_floats()
void prep_heads_args(const matcha_tensors& p, float* folded, float* merged, float* frag, PrepArgs& a) {
  const size_t wsz = (size_t)MATCHA_N_HEAD * 64 * 64, csz = (size_t)MATCHA_N_HEAD * 64;
  a.Wq = p.w_q; a.Wk = p.w_k; a.Wv = p.w_v;
  a.gq = p.ln_q_g; a.gk = p.ln_k_g; a.gv = p.ln_v_g; a.bq = p.ln_q_b; a.bv = p.ln_v_b;
  a.fc1_w = p.fc1_w; a.fc1_b = p.fc1_b; a.p0w = p.pff0_w; a.p0b = p.pff0_b; a.p1w = p.pff1_w; a.p1b = p.pff1_b;
  a.fwq = folded; a.fwk = folded + wsz; a.fwv = folded + 2 * wsz; a.fcq = folded + 3 * wsz; a.fcv = a.fcq + 2 * csz;
  const MergedView v = merged_view(merged);
  a.B = const_cast<float*>(v.B); a.M = const_cast<float*>(v.M); a.bvec = const_cast<float*>(v.bvec); a.bdyn = const_cast<float*>(v.bdyn);
  a.frag = reinterpret_cast<u32x4*>(frag);
}
int launch_prep_heads(const matcha_tensors& p, float* folded, float* merged, float* frag, hipStream_t st) {
  PrepArgs a;
  prep_heads_args(p, folded, merged, frag, a);
  hipLaunchKernelGGL(prep_heads_kernel, dim3(kPrepGridX, kPrepGridY, kPrepGridZ), dim3(256), 0, st, a);
  MATCHA_CHECK_LAUNCH("prep_heads_kernel");
  return MATCHA_OK;
}

int launch_fused_fwd32(const matcha_tensors& p, const float* folded, const float* frag, const float* X, const Ragged& rg, int64_t B, int L, const float* y,
                       const float* w, float* Y, float* H1, float* H2, float* logits, float* row_loss, const uint64_t* seed, float p_fc1, float p_pff,
                       hipStream_t st, float* ddyn0, float* dXs, float* tslab, float alpha, float* rimg, float* tail_dh2) {
  Fwd32Args g;
  g.tail_dh2 = nullptr;
  g.X = X; g.row_off = rg.row_off; g.tok_slot = rg.tok_slot; g.count = rg.count; g.half_meta = rg.half_meta; g.tok_pos = rg.tok_pos;
  g.L = L;
  g.wfrag = reinterpret_cast<const u32x4*>(frag);
  (void)folded;
  g.hp = HeadParams{p.pff_ln_g, p.pff_ln_b, p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b, p.cls_w, p.cls_b};
  g.y = y; g.w = w; g.Y = Y; g.H1 = H1; g.H2 = H2; g.logits = logits; g.row_loss = (y && w) ? row_loss : nullptr;
  g.seed = seed; g.p_fc1 = p_fc1; g.p_pff = p_pff;
  g.ddyn0 = (y && w) ? ddyn0 : nullptr; g.dXs = dXs; g.tslab = tslab; g.alpha_over_B = alpha / (float)B; g.qkv = rimg;
  size_t lds = ((size_t)2 * kHT + 64) * sizeof(float);
  auto launch = [&](auto kfn) { hipLaunchKernelGGL(kfn, dim3(rg.nhalves), dim3(64), lds, st, g); };
  // algorithmic flops per token (the reference formulation's): 8 heads x 4 GEMMs (Q, K, V, fc1 block) + the two pff GEMMs, 2*64*64 each
  ProfScope ps(MATCHA_PROF_FUSED_FWD, (double)(B * L + 1) * (MATCHA_N_HEAD * 4.0 + 2.0) * 2.0 * 64.0 * 64.0, st);
  const int ml = L <= 2 ? 2 : (L <= 6 ? L : 8);
  // small batches (at most two half tiles per CU even at the bound): the heads side by side in eight wavefronts per half tile
  if (fused_small_batch(rg)) {
    const size_t ldsh = ((size_t)10 * kHT + 64) * sizeof(float);
    auto launchh = [&](auto kfn) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh);
      hipLaunchKernelGGL(kfn, dim3(rg.nhalves), dim3(512), ldsh, st, g);
    };
    switch (ml) {
      case 2: launchh(fused_fwd32h_kernel<2>); break;
      case 3: launchh(fused_fwd32h_kernel<3>); break;
      case 4: launchh(fused_fwd32h_kernel<4>); break;
      case 5: launchh(fused_fwd32h_kernel<5>); break;
      case 6: launchh(fused_fwd32h_kernel<6>); break;
      default: launchh(fused_fwd32h_kernel<8>); break;
    }
    MATCHA_CHECK_LAUNCH("fused_fwd32h_kernel");
    return MATCHA_OK;
  }
  g.tail_dh2 = g.ddyn0 ? tail_dh2 : nullptr;
  switch (ml) {
    case 2: launch(fused_fwd32_kernel<2>); break;
    case 3: launch(fused_fwd32_kernel<3>); break;
    case 4: launch(fused_fwd32_kernel<4>); break;
    case 5: launch(fused_fwd32_kernel<5>); break;
    case 6: launch(fused_fwd32_kernel<6>); break;
    default: launch(fused_fwd32_kernel<8>); break;
  }
  MATCHA_CHECK_LAUNCH("fused_fwd32_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
