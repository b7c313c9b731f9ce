// Fused backward of the multi-head attention block for embed_dim 64 (training path of the metric's configuration), merged heads
// (per head r = B_h x_hat + b_h, s_ij = r_i . x_hat_j, z_i = sum_j p_ij x_hat_j, dyn += M_h z_i).
//
// Given X (the encoder input, [Tn, 64]), dDyn = dL/d(fc1 output before bias) ([Tn, 64], produced by the tail's backward inside the
// forward kernel or by the pff / tail backward kernels) and the record the training forward left per (half tile, head) -- its wavefront's
// r rows as register images + the attention probabilities --, fused_bwdh_kernel produces for ONE head
//   dB_h, dM_h, db_h, the column sums of dDyn, the padding token's d x_hat, and the head's contribution to d x_hat
// without materialising r / z / P or their gradients in HBM (Modules.py:519-572 backward).
//
// Work decomposition ("head-major"): workgroup (head, chunk) walks the half tiles of its chunk of hyperedges.  B_h and M_h stay in
// registers as MFMA fragments (three bf16 planes each, round 5), dB_h and dM_h accumulate in MFMA accumulators for the whole walk and are
// written once per workgroup into a slab.  fbm_reduce_kernel sums the slabs in a fixed order, fbm_chain_kernel applies the chain rule back to the folded projections
// (dW'q = W'k dB, dW'k = W'q dB^T + cq (x) db, dWfc1_h = dM W'v^T + ..., dW'v = Wfc1_h^T dM), fb_unfold_kernel un-folds the LayerNorm affines:
//   W' = W * g, c = W . b   =>   dW = dW' * g + dc (x) b,   dg = sum_n dW' * W,   db = W^T dc.
// The eight heads of a token add their d x_hat into one [T, 64] buffer (float atomics, one instruction = four token rows x 64 B) or, for
// deterministic / row-sparse callers, write eight slabs that lnhat_bwd_kernel / front_bwd_kernel sum in a fixed order.
//
// The reference's own four-product formulation (Q, K, V, fc1 per head) is NOT in this file any more: option disable_merged runs it on
// the layer-by-layer kernels (attention.hip, gemm_lds.hip), which is what the merged kernels are tested against.
#include <stdlib.h>
#include <string.h>

#include "bf16x3.hpp"
#include "kernels.hpp"
#include "tail_reduce.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kLd = 68;              // LDS row stride (floats)
constexpr float kEpsLn = 1e-5f;
constexpr int kWgSlab = 4 * 4096 + 6 * 64;   // dW'q dW'k dW'v dWfc1 | dcq dck dcv dKpad dVpad dfc1_b
constexpr int kVecOff = 4 * 4096;
constexpr int kMaxChunks = 64;

#define ZR8(p)                                                                                           \
  do {                                                                                                   \
    *reinterpret_cast<float4*>(p) = make_float4(0.f, 0.f, 0.f, 0.f);                                     \
    *reinterpret_cast<float4*>((p) + 4) = make_float4(0.f, 0.f, 0.f, 0.f);                               \
  } while (0)

__device__ __forceinline__ void ln_row16(const float4& v, float& mean, float& rstd) {
  const float s = group_sum<16>((v.x + v.y) + (v.z + v.w));
  mean = s * (1.f / 64.f);
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
  const float q = group_sum<16>((a * a + b * b) + (c * c + e * e));
  rstd = __builtin_amdgcn_rsqf(q * (1.f / 64.f) + kEpsLn);
}

// Attention forward + backward on LDS tiles, token-parallel: one 8-lane group per QUERY token (lane `sub` owns features
// [8 sub, 8 sub + 8)), so a tile's <= 63 tokens fill the 32 groups of the workgroup in two passes and the work per group
// is O(k) instead of O(k^2).  Formulas: attention.hip (attn_bwd_kernel).
//   row phase (token i):    P_i., Pp_i, O_i, dS_i., dSp_i, dQ_i  -- everything that needs only row i of the score matrix;
//                           P_i. and dS_i. go to LDS ([64][8] each), O_i and dQ_i stay in registers
//   column phase (token j): dV_j = sum_i P_ij dO_i,  dK_j = sum_i dS_ij Q_i   over the hyperedge's rows
//   write phase:            O -> Fs, dQ/dK/dV over Q/K/V (after a barrier: the column phase reads Q and dO of other rows)
// acc: this lane's running sums {dK_pad, dV_pad} x 8 features.
// Eight consecutive floats as four packed pairs: dot products and axpys compile to v_pk_mul / v_pk_fma (two flops per lane
// per instruction) instead of scalar chains the compiler re-packs with extra moves.

// The attention rows are STREAMED: one key / value row in use and one in flight.  FB_PIN(addr, x): an empty asm that takes the LDS offset
// of the NEXT row and the eight registers of the running sum as in / out operands: the load of row j + 1 cannot be issued before the
// arithmetic on row j - 1 has produced x.  (sched_barrier alone does not do it: instruction selection has already placed the unchained
// LDS loads of all rows ahead of the arithmetic -- 40 to 80 operand registers -- when the machine scheduler sees the fence.)
#define FB_PIN(addr, x) asm volatile("" : "+v"(addr), "+v"((x).a), "+v"((x).b), "+v"((x).c), "+v"((x).d))
// Keys = values (merged heads: both are the x_hat rows, kpad = vpad = the padding token's x_hat): ONE pass over the rows.  With the weights
// w_j = p_ij (w_pad = n_pad p_i,pad) and d_j = dz_i . x_j:   z_i = sum_j w_j x_j,   sig = sum_j w_j d_j,   A = sum_j (w_j d_j) x_j, and
//   d r_i = sum_j dS_ij x_j = (A - sig z_i) / temp        (dS_ij = w_j (d_j - sig) / temp; the sums include the padding term),
// so the second read of every row (attn_row8's K pass) and its address arithmetic go away.  Same outputs as attn_row8.
template <int ML>
__device__ __forceinline__ void attn_row8_kv(const float* __restrict__ Qs, const float* __restrict__ Xs, const float* __restrict__ Fs,
                                             const float* __restrict__ xpad, const float* __restrict__ Ps, float* __restrict__ dSs, int li, int li0,
                                             int k, int n_pad, int sub, float inv_temp, V8& o, V8& gq, V8& accK, V8& accV) {
  const float padf = (float)n_pad;
  const bool hp = n_pad > 0;
  float p[ML], ds[ML], pp;
  {
    const float4 pa = *reinterpret_cast<const float4*>(&Ps[li * 8]), pb = *reinterpret_cast<const float4*>(&Ps[li * 8 + 4]);
    const float w[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
    for (int j = 0; j < ML; ++j) p[j] = (j < k) ? w[j] : 0.f;
    pp = hp ? w[7] : 0.f;
  }
  const float ppf = padf * pp;
  const int ro0 = li0 * kLd + 8 * sub;                   // row j (clamped to the hyperedge): ro0 + min(j, k - 1) rows
  const V8 go = ld8(&Fs[li * kLd + 8 * sub]);
  const V8 vp = ld8(xpad + 8 * sub);
  const float dsp_raw = group_sum8_dpp(dot8(go, vp));
  float sig = ppf * dsp_raw;
  o = scale8(ppf, vp);
  V8 a = scale8(ppf * dsp_raw, vp);
  V8 vn = ld8(&Xs[ro0]), vn2 = ld8(&Xs[ro0 + (1 < k ? 1 : 0) * kLd]);      // two rows in flight
#pragma unroll
  for (int j = 0; j < ML; ++j) {
    const V8 v = vn;
    vn = vn2;
    if (j + 2 < ML) {
      int ad = ro0 + (j + 2 < k ? j + 2 : 0) * kLd;
      FB_PIN(ad, o);
      vn2 = ld8(&Xs[ad]);
    }
    axpy8(o, p[j], v);
    const float d = group_sum8_dpp(dot8(go, v));
    ds[j] = d;
    const float wd = p[j] * d;
    sig += wd;
    axpy8(a, wd, v);
  }
#pragma unroll
  for (int j = 0; j < ML; ++j) ds[j] = p[j] * (ds[j] - sig) * inv_temp;
  const float dspf = padf * (pp * (dsp_raw - sig) * inv_temp);
  axpy8(accV, ppf, go);
  // d r_i = (A - sig z_i) / temp
  {
    const float ns = -sig;
    axpy8(a, ns, o);
    gq = scale8(inv_temp, a);
  }
  {
    int ad = li * kLd + 8 * sub;
    FB_PIN(ad, gq);
    const V8 q = ld8(&Qs[ad]);
    axpy8(accK, dspf, q);
  }
  if (sub == 1) {                                          // row i of dS for the column phase (P is in Ps already)
    float* dst = dSs + li * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(ds[0], ds[1 % ML], ML > 2 ? ds[2 % ML] : 0.f, ML > 3 ? ds[3 % ML] : 0.f);
    if (ML > 4) *reinterpret_cast<float4*>(dst + 4) = make_float4(ds[4 % ML], ML > 5 ? ds[5 % ML] : 0.f, ML > 6 ? ds[6 % ML] : 0.f, ML > 7 ? ds[7 % ML] : 0.f);
  }
}

template <int ML>
__device__ __forceinline__ void attn_col8(const float* __restrict__ Qs, const float* __restrict__ Fs, const float* __restrict__ Ps,
                                          const float* __restrict__ dSs, int li, int li0, int k, int sub, V8& gk, V8& gv) {
  const int jj = li - li0;
  gk = zero8();
  gv = zero8();
  const int ro0 = li0 * kLd + 8 * sub;
  V8 qn = ld8(&Qs[ro0]), gn = ld8(&Fs[ro0]);
#pragma unroll
  for (int i = 0; i < ML; ++i) {
    const V8 q = qn, go = gn;
    const int ri = li0 + (i < k ? i : 0);
    // unconditional loads (the row index is clamped into the hyperedge, whose rows hold finite values) + a select: `cond ? load : 0` becomes a
    // branch around the load
    const float pl = Ps[ri * 8 + jj], dl = dSs[ri * 8 + jj];
    const float pij = (i < k) ? pl : 0.f, dsij = (i < k) ? dl : 0.f;
    if (i + 1 < ML) {
      int a = ro0 + (i + 1 < k ? i + 1 : 0) * kLd;
      FB_PIN(a, gk);
      qn = ld8(&Qs[a]); gn = ld8(&Fs[a]);
    }
    axpy8(gv, pij, go);
    axpy8(gk, dsij, q);
  }
}

// -DFB_TIMING: per-phase wall-clock (100 MHz) of workgroup 0, printed at the end -- development builds only
#ifdef FB_TIMING
#define FB_T(i) do { const long long now__ = wall_clock64(); tph[i] += now__ - tlast; tlast = now__; } while (0)
#else
#define FB_T(i) do { } while (0)
#endif

#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)      /* the small per-step products (fbm_chain_kernel) */

// ---- the backward of  r = B_h x + b_h,  s_ij = r_i . x_j,  z_i = sum_j p_ij x_j,  dyn += M_h z  (merged heads) ----------
// Per (half tile, head) FOUR 64 x 64 products instead of the reference formulation's eight:  dZ = dDyn M_h;  d x_hat = dR B_h + (the
// attention's own gradient into its keys / values, which ARE the x_hat rows);  dB_h += dR^T x_hat;  dM_h += dDyn^T Z.  The attention runs
// attn_row8_kv / attn_col8 with Q := r, K := V := x_hat, dO := dZ; what those return as dK + dV is d x_hat's attention part (tile Gs), what
// they accumulate for the padding key / value is d x_hat of the padding token.  fbm_chain_kernel then applies the chain rule from
// (dB_h, db_h, dM_h, d bdyn) to the folded projections, and the LayerNorm un-folding follows.
constexpr int kWgSlabM = 2 * 4096 + 4 * 64;     // dB_h dM_h | db_h (column sums of dR), d bdyn (column sums of dDyn), dxpad, spare
constexpr int kVecOffM = 2 * 4096;
// Walk: workgroup = (head, chunk of the forward's own HALF tiles: whole hyperedges, <= 31 tokens, ragged.hip half_meta), FOUR wavefronts,
// TWO workgroups per CU -- at different phases, so one's attention (latency-bound vector + LDS work) runs under the other's GEMMs.  Every
// wavefront owns 16 feature columns of each product.  The forward's saved record is per (half tile, head): its wavefront's register image.
constexpr int kTileH = 32 * kLd;
// ---- round 5: the four products on the bf16 matrix pipe (fp32-accurate: three bf16 planes per operand, six plane products; the note is in
// fused_fwd32.hip, the arithmetic restated in tests/test_cpu_bf16x3.py) ---------------------------------------------------------------------
// The f32 MFMA shares the vector ALUs with the attention arithmetic (128 x 32 cycles per wavefront and half tile next to ~600 vector
// instructions); v_mfma_f32_16x16x32_bf16 has its own pipe.  What makes it pay HERE is that every GEMM operand is split ONCE where it is
// produced and kept in LDS as bf16 planes -- dDyn and x_hat by the staging threads (one split per workgroup, not per wavefront: a 16 x 16
// output tile reuses a fragment only six times), dR and Z by the attention's write phase, B_h and M_h once per workgroup walk in registers:
//   plane tile = [32 tokens][80 bf16] x 3 planes (160-byte rows: with 144-byte rows a third of the LDS cycles were bank conflicts --
//     SQ_LDS_BANK_CONFLICT 70 M of SQ_LDS_IDX_ACTIVE 221 M per launch; on 40-dword rows both kinds of read are conflict-free)
//   row fragment (contraction over FEATURES: dZ^T = M_h^T dDyn^T, d x_hat = dR B_h)   = one ds_read_b128 per plane
//   column fragment (contraction over TOKENS: dB_h += dR^T x_hat, dM_h += dDyn^T Z)   = two ds_read_b64_tr_b16 per plane -- the hardware
//     transpose read hands lane i of a 16-lane group column i of a 4 row x 16 column block (MI355X guide T10)
// LDS per workgroup (81 280 B: two workgroups per CU just fit): x_hat f32 (the attention's keys / values) + planes, dDyn planes, RB = r f32 ->
// dR planes, FB = dZ f32 -> Z planes, G f32.  One set only: the next half tile's rows wait in registers (fetched during the GEMMs) and are
// staged at the top of the next iteration.
constexpr int kPS = 80;                 // bf16 per plane row: 40 dwords -- with the token slots below BOTH fragment reads are conflict-free (bank sets
                                        // {0,40,16,56,32,8,48,24} + 4 kq for the 16-byte row reads, 8-bank slots 40 r mod 64 for eight consecutive rows)
constexpr int kPlane = 32 * kPS;        // bf16 per plane
constexpr int kPT = 3 * kPlane;         // bf16 per three-plane tile (15 360 B)
// column fragments: contraction slot 8 kq + j holds token 4 kq + j (j < 4) or 16 + 4 kq + (j - 4) -- free, both operands of a product use the
// same map -- so that a 32-lane half reads EIGHT CONSECUTIVE rows per transposed read.  The lane's block address: row 4 kq + ((lane & 15) >> 2),
// columns c0 + 4 (lane & 3); lane i of the 16-lane group receives column c0 + i
typedef Planes<kPS, 16> PL;
__device__ __forceinline__ Frag3 frag_row(const short* __restrict__ p) { return PL::row(p); }
__device__ __forceinline__ void frag_store(short* __restrict__ p, const Frag3& f) { PL::store(p, f); }
__device__ __forceinline__ Frag3 frag_col(const short* __restrict__ p) { return PL::col(p); }

struct FusedBwdHArgs {
  const float* X; const float* dDyn; const int32_t* count; const int32_t* half_meta; const int32_t* tok_pos;
  int L; int nhalves; int nchunks;
  const float* mB; const float* mM;               // merged matrices [8][64][64] (launch_prep_heads)
  float* dxh; int64_t tcap;
  int dx_atomic;                                   // 1: every head adds into ONE [tcap][64] buffer with float atomics (zeroed by the launcher); 0: one slab per head
  float* wslab;                                    // [8][nchunks][kWgSlabM]
  const float* rimg;                               // [nhalves][8][kImgRecH]: r rows (register images) + attention probabilities of the forward
};
constexpr size_t kBwdLdsBytes = (size_t)2 * kTileH * 4 + (size_t)4 * kPT * 2 + (64 + 256 + 256 + 32) * 4;

template <int ML>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fused_bwdh_kernel(FusedBwdHArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                                   // x_hat f32 (keys = values of the attention)
  float* Gs = lds + kTileH;                          // attention's gradient into the x_hat rows (keys + values)
  short* Xp = reinterpret_cast<short*>(lds + 2 * kTileH);   // x_hat planes
  short* Dp = Xp + kPT;                              // dDyn planes
  short* RBp = Dp + kPT;                             // r f32 -> dR planes
  short* FBp = RBp + kPT;                            // dZ f32 -> Z planes
  float* Rs = reinterpret_cast<float*>(RBp);
  float* Fs = reinterpret_cast<float*>(FBp);
  float* sm = reinterpret_cast<float*>(FBp + kPT);
  float* xpad = sm;
  float* dSs = xpad + 64;             // [32][8]
  float* Ps = dSs + 256;              // [32][8]
  int* tinfo = reinterpret_cast<int*>(Ps + 256);     // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kq = lane >> 4;           // 16x16x32 fragments: row / column c16, contraction slots 8 kq + {0..7}
  const int fb = 16 * wave;                            // this wave's 16 feature columns
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;     // staging: 16 rows x 16 lanes (float4), twice

  int head, chunk;
  if ((g.nchunks & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    head = j & 7;
    chunk = (j >> 3) * 8 + xcd;
  } else {
    head = blockIdx.x & 7;
    chunk = blockIdx.x >> 3;
  }
  const int tr = g.count[1];
  int nh = g.count[3];
  if (nh > g.nhalves) nh = g.nhalves;
  const int per = (nh + g.nchunks - 1) / g.nchunks;
  const int tile_lo = chunk * per;
  const int tile_hi = (tile_lo + per < nh) ? tile_lo + per : nh;
  const float inv_temp = 0.125f;
#ifdef FB_TIMING
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = wall_clock64();
#endif

  if (tid < 16) {                     // the padding token's x_hat
    const float4 xv = *reinterpret_cast<const float4*>(g.X + (int64_t)tr * 64 + sc4);
    float m, rs;
    ln_row16(xv, m, rs);
    *reinterpret_cast<float4*>(&xpad[sc4]) = make_float4((xv.x - m) * rs, (xv.y - m) * rs, (xv.z - m) * rs, (xv.w - m) * rs);
  }

  // weight-gradient accumulators: rows 16 mt + 4 kq + reg, column fb + c16 of dB_h and dM_h (four 16 x 16 tiles each)
  f32x4 ab0 = {0.f, 0.f, 0.f, 0.f}, ab1 = ab0, ab2 = ab0, ab3 = ab0, am0 = ab0, am1 = ab0, am2 = ab0, am3 = ab0;
  V8 accK = zero8(), accV = zero8();                  // the padding token's gradient as a key (sum dS_i,pad r_i) and as a value (sum p_i,pad dz_i)
  // column sums of dDyn (columns sc4 + {0..3} over the rows this thread stages) and of dR (features 8 sub + {0..7} over this lane group's tokens)
  f2 cd0 = {0.f, 0.f}, cd1 = cd0;
  V8 accR = zero8();

  const int4* meta = reinterpret_cast<const int4*>(g.half_meta);
  const int4 mzero = make_int4(0, 0, 0, 0);
  int4 mc = tile_lo < tile_hi ? meta[tile_lo] : mzero;
  int4 mn = tile_lo + 1 < tile_hi ? meta[tile_lo + 1] : mzero;
  float4 xn0, xn1, dn0, dn1;
  int tpn = 0;
  f32x4 ri0, ri1, pn = {0.f, 0.f, 0.f, 0.f};
#define FBH_ROW_GLOAD(I, M)                                                                              \
  do {                                                                                                   \
    const int row__ = srow + 16 * (I);                                                                   \
    const int64_t tok__ = (M).x + (row__ < (M).y ? row__ : ((M).y > 0 ? (M).y - 1 : 0));                 \
    xn##I = *reinterpret_cast<const float4*>(g.X + tok__ * 64 + sc4);                                    \
    dn##I = *reinterpret_cast<const float4*>(g.dDyn + tok__ * 64 + sc4);                                 \
  } while (0)
#define FBH_ROWS_GLOAD(M)                                                                                \
  do {                                                                                                   \
    FBH_ROW_GLOAD(0, M); FBH_ROW_GLOAD(1, M);                                                            \
    if (tid < 32) tpn = g.tok_pos[(M).x + (tid < (M).y ? tid : ((M).y > 0 ? (M).y - 1 : 0))];           \
  } while (0)
  // x_hat row -> Xs (f32) and Xp (planes); dDyn row (zero past the tokens) -> Dp (planes) + its column sums
#define FBH_ROW_STAGE(I)                                                                                 \
  do {                                                                                                   \
    const int row__ = srow + 16 * (I);                                                                   \
    const float msk__ = row__ < n_real ? 1.f : 0.f;                                                      \
    const float4 xv__ = xn##I, dv__ = dn##I;                                                             \
    const float mean__ = group_sum16_dpp((xv__.x + xv__.y) + (xv__.z + xv__.w)) * (1.f / 64.f);          \
    const float a__ = xv__.x - mean__, b__ = xv__.y - mean__, c__ = xv__.z - mean__, e__ = xv__.w - mean__; \
    const float q__ = group_sum16_dpp((a__ * a__ + b__ * b__) + (c__ * c__ + e__ * e__));                \
    const float rs__ = msk__ * __builtin_amdgcn_rsqf(q__ * (1.f / 64.f) + kEpsLn);                       \
    const float4 xh__ = make_float4(a__ * rs__, b__ * rs__, c__ * rs__, e__ * rs__);                     \
    *reinterpret_cast<float4*>(&Xs[row__ * kLd + sc4]) = xh__;                                           \
    {                                                                                                    \
      const P3 p0__ = split2(xh__.x, xh__.y), p1__ = split2(xh__.z, xh__.w);                             \
      short* d__ = Xp + row__ * kPS + sc4;                                                               \
      *reinterpret_cast<u32x2*>(d__) = (u32x2){p0__.h, p1__.h}; *reinterpret_cast<u32x2*>(d__ + kPlane) = (u32x2){p0__.m, p1__.m}; \
      *reinterpret_cast<u32x2*>(d__ + 2 * kPlane) = (u32x2){p0__.l, p1__.l};                             \
    }                                                                                                    \
    const float4 dm__ = make_float4(dv__.x * msk__, dv__.y * msk__, dv__.z * msk__, dv__.w * msk__);     \
    {                                                                                                    \
      const P3 p0__ = split2(dm__.x, dm__.y), p1__ = split2(dm__.z, dm__.w);                             \
      short* d__ = Dp + row__ * kPS + sc4;                                                               \
      *reinterpret_cast<u32x2*>(d__) = (u32x2){p0__.h, p1__.h}; *reinterpret_cast<u32x2*>(d__ + kPlane) = (u32x2){p0__.m, p1__.m}; \
      *reinterpret_cast<u32x2*>(d__ + 2 * kPlane) = (u32x2){p0__.l, p1__.l};                             \
    }                                                                                                    \
    cd0 += (f2){dm__.x, dm__.y}; cd1 += (f2){dm__.z, dm__.w};                                            \
  } while (0)
  // the forward wavefront's register image: float4 index (wc * 4 + gq) * 64 + 32 h + r holds features 32 wc + 8 gq + 4 h + {0..3} of row r
#define FBH_RIMG_GLOAD(HALF)                                                                             \
  do {                                                                                                   \
    const f32x4* r__ = reinterpret_cast<const f32x4*>(g.rimg + ((int64_t)(HALF) * MATCHA_N_HEAD + head) * kImgRecH);  \
    if (tid < 64) pn = __builtin_nontemporal_load(r__ + 512 + tid);                                      \
    ri0 = __builtin_nontemporal_load(r__ + tid); ri1 = __builtin_nontemporal_load(r__ + 256 + tid);      \
  } while (0)
  // B_h and M_h as register fragments for the whole walk: lane (c16, kq), step s holds W[32 s + 8 kq + {0..7}][fb + c16] -- the A operand of
  // dZ^T = M_h^T dDyn^T (rows = features) and the B operand of d x_hat = dR B_h (columns = features); three planes each
  Frag3 Mf[2], Bf[2];
  {
    FBH_ROWS_GLOAD(mc);
    if (tile_lo < tile_hi) FBH_RIMG_GLOAD(tile_lo);
    const float* mp = g.mM + (int64_t)head * 4096 + (8 * kq) * 64 + fb + c16;
    const float* bp = g.mB + (int64_t)head * 4096 + (8 * kq) * 64 + fb + c16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float vm[8], vb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { vm[j] = mp[(32 * s + j) * 64]; vb[j] = bp[(32 * s + j) * 64]; }
      Mf[s] = split8(vm); Bf[s] = split8(vb);
    }
  }

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int4 mnn = tile + 2 < tile_hi ? meta[tile + 2] : mzero;
    const int t0 = mc.x, n_real = mc.y;
    FB_T(0);
    __syncthreads();                                  // the previous half tile's GEMMs are done with every tile
    FB_T(7);
    // ---- stage this half tile (its rows were fetched during the previous one's GEMMs) ----
    {
      FBH_ROW_STAGE(0); FBH_ROW_STAGE(1);
      if (tid < 32) tinfo[tid] = tid < n_real ? ((tid - (tpn & 255)) | (tpn & ~255)) : 0;
      f32x4* d__ = reinterpret_cast<f32x4*>(&Rs[(lane & 31) * kLd + 8 * wave + 4 * (lane >> 5)]);
      d__[0] = ri0; d__[8] = ri1;
      if (tid < 64) reinterpret_cast<f32x4*>(Ps)[tid] = pn;
    }
    __syncthreads();
    // per-lane indices re-derived from an opaque copy of the thread id (loop-invariant addresses are hoisted and spilled otherwise)
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int c16 = lane & 15, kq = lane >> 4;
    const int fb = 16 * wave;
    const int srow = tid_ >> 4, sc4 = (tid_ & 15) * 4;
    const int sub = lane & 7;
    (void)srow; (void)sc4;
    // ---- dZ^T = M_h^T . dDyn^T: lane (c16, kq) ends with token c16 (+ 16) and features fb + 4 kq + {0..3} ----
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const short* dp = Dp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag3 b0 = frag_row(dp + 32 * s), b1 = frag_row(dp + 16 * kPS + 32 * s);
        acc0 = mma6(acc0, Mf[s], b0); acc1 = mma6(acc1, Mf[s], b1);
      }
      *reinterpret_cast<f32x4*>(&Fs[c16 * kLd + fb + 4 * kq]) = acc0;
      *reinterpret_cast<f32x4*>(&Fs[(16 + c16) * kLd + fb + 4 * kq]) = acc1;
    }
    FB_T(1);
    __syncthreads();
    FB_T(7);
    // ---- attention forward + backward in x_hat space: 8 lanes per token, all 32 rows in one pass ----
    {
      V8 o0, q0, k0, v0;
      const int la = wave * 8 + (lane >> 3);
      const bool acta = la < n_real;
      int ia = 0;
      if (acta) { ia = tinfo[la]; attn_row8_kv<ML>(Rs, Xs, Fs, xpad, Ps, dSs, la, ia & 255, ia >> 8, g.L - (ia >> 8), sub, inv_temp, o0, q0, accK, accV); }
      __builtin_amdgcn_sched_barrier(0);
      FB_T(2);
      __syncthreads();
      FB_T(7);
      if (acta) {
        attn_col8<ML>(Rs, Fs, Ps, dSs, la, ia & 255, ia >> 8, sub, k0, v0);
        k0.a += v0.a; k0.b += v0.b; k0.c += v0.c; k0.d += v0.d;      // the row is key AND value: d x_hat_j = sum_i dS_ij r_i + p_ij dz_i
        st8(&Gs[la * kLd + 8 * sub], k0);
      } else {
        ZR8(&Gs[la * kLd + 8 * sub]);
      }
      FB_T(3);
      __syncthreads();                                // every column phase is done with the r and dZ rows: they become the dR and Z PLANES
      FB_T(7);
      if (!acta) { o0 = zero8(); q0 = zero8(); }      // rows past the tokens: zero planes (they are contraction slots of the weight gradients)
      frag_store(FBp + la * kPS + 8 * sub, split8(o0));
      frag_store(RBp + la * kPS + 8 * sub, split8(q0));
      accR.a += q0.a; accR.b += q0.b; accR.c += q0.c; accR.d += q0.d;
    }
    FB_T(4);
    __syncthreads();
    FB_T(7);
    FBH_ROWS_GLOAD(mn);                               // next half tile's rows: in flight during the GEMMs below
    // ---- this head's share of d x_hat = dR B_h + Gs ----
    if (g.dx_atomic) {
      // rows = tokens 4 kq + reg (+ 16), columns = features fb + c16: one atomic instruction covers 4 token rows x 64 contiguous bytes.  The
      // eight heads of a chunk run on the same XCD at about the same time: the adds meet in that L2, and d x_hat leaves it once
      const float* gp = Gs + (4 * kq) * kLd + fb + c16;
      f32x4 dx0 = {gp[0], gp[kLd], gp[2 * kLd], gp[3 * kLd]};
      f32x4 dx1 = {gp[16 * kLd], gp[17 * kLd], gp[18 * kLd], gp[19 * kLd]};
      const short* arow = RBp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag3 a0 = frag_row(arow + 32 * s), a1 = frag_row(arow + 16 * kPS + 32 * s);
        dx0 = mma6(dx0, a0, Bf[s]); dx1 = mma6(dx1, a1, Bf[s]);
      }
      float* out = g.dxh + ((int64_t)t0 + 4 * kq) * 64 + fb + c16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        if (4 * kq + reg < n_real) unsafeAtomicAdd(out + reg * 64, dx0[reg]);
        if (16 + 4 * kq + reg < n_real) unsafeAtomicAdd(out + (16 + reg) * 64, dx1[reg]);
      }
    } else {
      f32x4 dx0 = *reinterpret_cast<const f32x4*>(&Gs[c16 * kLd + fb + 4 * kq]);
      f32x4 dx1 = *reinterpret_cast<const f32x4*>(&Gs[(16 + c16) * kLd + fb + 4 * kq]);
      const short* arow = RBp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag3 a0 = frag_row(arow + 32 * s), a1 = frag_row(arow + 16 * kPS + 32 * s);
        dx0 = mma6(dx0, Bf[s], a0); dx1 = mma6(dx1, Bf[s], a1);
      }
      float* out = g.dxh + ((int64_t)head * g.tcap + t0 + c16) * 64 + fb + 4 * kq;
      if (c16 < n_real) *reinterpret_cast<f32x4*>(out) = dx0;
      if (16 + c16 < n_real) *reinterpret_cast<f32x4*>(out + 16 * 64) = dx1;
    }
    FB_T(5);
    if (tile + 1 < tile_hi) FBH_RIMG_GLOAD(tile + 1);     // next half tile's r rows and probabilities: in flight during the weight-gradient GEMMs
    // ---- weight gradients: dB[a][b] += sum_t dR[t][a] x_hat[t][b];  dM[n][m] += sum_t dDyn[t][n] Z[t][m]: ONE 32-token step, column fragments ----
    {
      const int blk = ((4 * kq + ((lane & 15) >> 2)) * kPS) + 4 * (lane & 3);      // this lane's address inside a 4 x 16 transpose block
      const Frag3 xb = frag_col(Xp + blk + fb), zb = frag_col(FBp + blk + fb);
      {
        const Frag3 r0 = frag_col(RBp + blk), r1 = frag_col(RBp + blk + 16);
        ab0 = mma6(ab0, r0, xb); ab1 = mma6(ab1, r1, xb);
        const Frag3 r2 = frag_col(RBp + blk + 32), r3 = frag_col(RBp + blk + 48);
        ab2 = mma6(ab2, r2, xb); ab3 = mma6(ab3, r3, xb);
      }
      {
        const Frag3 d0 = frag_col(Dp + blk), d1 = frag_col(Dp + blk + 16);
        am0 = mma6(am0, d0, zb); am1 = mma6(am1, d1, zb);
        const Frag3 d2 = frag_col(Dp + blk + 32), d3 = frag_col(Dp + blk + 48);
        am2 = mma6(am2, d2, zb); am3 = mma6(am3, d3, zb);
      }
    }
    FB_T(6);
    mc = mn; mn = mnn;
  }

  // ---- workgroup slab ----
  __syncthreads();
#ifdef FB_TIMING
  if (blockIdx.x == 0 && (tid == 0 || tid == 192))
    printf("fused_bwdh wg0 wave %d us: stage %.1f dZ %.1f attn-row %.1f attn-col %.1f attn-write %.1f dx %.1f tn %.1f barrier-wait %.1f (halves %d)\n", tid >> 6,
           tph[0] * 0.01, tph[1] * 0.01, tph[2] * 0.01, tph[3] * 0.01, tph[4] * 0.01, tph[5] * 0.01, tph[6] * 0.01, tph[7] * 0.01, tile_hi - tile_lo);
#endif
  float* slab = g.wslab + ((int64_t)head * g.nchunks + chunk) * kWgSlabM;
  {
    const int col = fb + c16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = 4 * kq + reg;
      slab[0 * 4096 + row * 64 + col] = ab0[reg]; slab[0 * 4096 + (row + 16) * 64 + col] = ab1[reg];
      slab[0 * 4096 + (row + 32) * 64 + col] = ab2[reg]; slab[0 * 4096 + (row + 48) * 64 + col] = ab3[reg];
      slab[1 * 4096 + row * 64 + col] = am0[reg]; slab[1 * 4096 + (row + 16) * 64 + col] = am1[reg];
      slab[1 * 4096 + (row + 32) * 64 + col] = am2[reg]; slab[1 * 4096 + (row + 48) * 64 + col] = am3[reg];
    }
  }
  // column sums of dDyn: thread (srow, sc4) staged rows srow, srow + 16 of every half tile -- the four row groups of a wave in a fixed xor order,
  // then the four waves in order
  float* redd = lds;                    // [4][64]
  {
    const float cdv[4] = {cd0.x, cd0.y, cd1.x, cd1.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = cdv[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) redd[wave * 64 + 4 * lane + i] = v;
    }
  }
  // column sums of dR and d x_hat of the padding token: the 8 lanes with equal `sub` of a wave (fixed xor tree), then the 4 waves in order
  float* redr = lds + 4 * 64;           // [4][64]
  float* redp = lds + 8 * 64;           // [4][64]
  const float accp[16] = {accK.a.x + accV.a.x, accK.a.y + accV.a.y, accK.b.x + accV.b.x, accK.b.y + accV.b.y,
                          accK.c.x + accV.c.x, accK.c.y + accV.c.y, accK.d.x + accV.d.x, accK.d.y + accV.d.y,
                          accR.a.x, accR.a.y, accR.b.x, accR.b.y, accR.c.x, accR.c.y, accR.d.x, accR.d.y};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = accp[i];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 8) (i < 8 ? redp : redr)[wave * 64 + 8 * lane + (i & 7)] = v;
  }
  __syncthreads();
  if (tid < 64) {
    slab[kVecOffM + 64 + tid] = (redd[tid] + redd[64 + tid]) + (redd[128 + tid] + redd[192 + tid]);    // d bdyn partial
    slab[kVecOffM + tid] = (redr[tid] + redr[64 + tid]) + (redr[128 + tid] + redr[192 + tid]);         // db_h partial
    slab[kVecOffM + 128 + tid] = (redp[tid] + redp[64 + tid]) + (redp[128 + tid] + redp[192 + tid]);   // dxpad partial
  }
}

// Chain rule from the merged matrices back to the folded projections, one block per head: sums the chunk slabs in order, then
//   B_h = W'k^T W'q, b_h = W'k^T cq:   dW'q = W'k dB;   dW'k = W'q dB^T + cq (x) db;   dcq = W'k db;   dck = 0
//   M_h = Wf_h W'v, bdyn = fc1_b + sum_h Wf_h cv_h:   dWf_h = dM W'v^T + dbdyn (x) cv_h;   dW'v = Wf_h^T dM;   dcv_h = Wf_h^T dbdyn
// and writes ONE slab per head in fused_bwd8_kernel's format (dW'q dW'k dW'v dWfc1 | dcq dck dcv 0 0 dfc1_b), which fb_unfold_kernel /
// fb_unfold2_kernel turn into the gradients of the original parameters as before (nchunks = 1); dxpad is summed here.
struct ChainArgs {
  const float* wslab; int nchunks;           // merged slabs [8][nchunks][kWgSlabM]
  float* red;                                // [8][kWgSlabM] chunk sums (fbm_reduce_kernel)
  const float* wq; const float* wk; const float* wv; const float* cq; const float* cv;   // folded
  const float* fc1_w;
  float* out;                                // [8][kWgSlab]
  float* dxpad;                              // [64]
};
// chunk sums of every slab element in chunk order; dxpad = the sum over heads and chunks of the padding token's share.
// Block roles: blocks [0, kFbmReduceBlocks) sum the backward kernel's slabs; the blocks behind them sum the slabs of the TAIL's parameter
// gradients that the forward left (tail_reduce.hpp) -- independent sums that used to be two launches, one in front of the backward kernel and
// one behind it.
constexpr int kFbmReduceX = (kWgSlabM + 255) / 256;
constexpr int kFbmReduceBlocks = kFbmReduceX * MATCHA_N_HEAD;
__global__ __launch_bounds__(256) void fbm_reduce_kernel(ChainArgs a, TailReduceArgs tl, int tail_blocks) {
  __shared__ float4 part[8 * 32];
  if ((int)blockIdx.x >= kFbmReduceBlocks) {
    tail_reduce_role(tl, (int)blockIdx.x - kFbmReduceBlocks, part);
    return;
  }
  const int head = blockIdx.x / kFbmReduceX, i = (blockIdx.x % kFbmReduceX) * 256 + threadIdx.x;
  if (i >= kWgSlabM) return;
  const float* base = a.wslab + (int64_t)head * a.nchunks * kWgSlabM + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int c = 0;
  for (; c + 7 < a.nchunks; c += 8) {                  // eight chunk slabs in flight per thread
    s0 += base[(int64_t)c * kWgSlabM]; s1 += base[(int64_t)(c + 1) * kWgSlabM]; s2 += base[(int64_t)(c + 2) * kWgSlabM]; s3 += base[(int64_t)(c + 3) * kWgSlabM];
    s4 += base[(int64_t)(c + 4) * kWgSlabM]; s5 += base[(int64_t)(c + 5) * kWgSlabM]; s6 += base[(int64_t)(c + 6) * kWgSlabM]; s7 += base[(int64_t)(c + 7) * kWgSlabM];
  }
  for (; c < a.nchunks; ++c) s0 += base[(int64_t)c * kWgSlabM];
  a.red[(int64_t)head * kWgSlabM + i] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}
// out[i][j] (16 rows of a 64 x 64 product per block) = sum_x A(i, x) B(x, j) [+ u[i] v[j]] with strided operands staged through LDS
__device__ __forceinline__ void mm64_slice(const float* __restrict__ A, int a_rs, int a_cs, const float* __restrict__ B, int b_rs, int b_cs,
                                           const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ out, int slice,
                                           float* __restrict__ As, float* __restrict__ Bs) {
  const int tid = threadIdx.x;
  {
    // every load of the two operand tiles in flight before the first LDS store (strided operands: scalar loads)
    float av[4], bv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int i = tid + 256 * t; av[t] = A[(int64_t)(16 * slice + (i >> 6)) * a_rs + (int64_t)(i & 63) * a_cs]; }
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int i = tid + 256 * t; bv[t] = B[(int64_t)(i >> 6) * b_rs + (int64_t)(i & 63) * b_cs]; }
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int i = tid + 256 * t; As[(i >> 6) * 65 + (i & 63)] = av[t]; }
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int i = tid + 256 * t; Bs[(i >> 6) * 65 + (i & 63)] = bv[t]; }
  }
  __syncthreads();
  // 16 x 64 outputs = one 16 x 16 tile per wavefront, 16 steps of v_mfma_f32_16x16x4_f32 over the contraction index
  const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = MFMA16(As[c16 * 65 + 4 * kk + kq], Bs[(4 * kk + kq) * 65 + 16 * wave + c16], acc);
  const int j = 16 * wave + c16;
  const float vj = u ? v[j] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int i = 4 * kq + reg;
    out[(16 * slice + i) * 64 + j] = acc[reg] + (u ? u[16 * slice + i] * vj : 0.f);
  }
}
// grid (4 row slices, 4 jobs, 8 heads): the four matrix gradients of a head in fused_bwd8_kernel's slab format; slice 0 also the vectors
__global__ __launch_bounds__(256) void fbm_chain_kernel(ChainArgs a) {
  __shared__ float As[16 * 65];
  __shared__ float Bs[64 * 65];
  const int slice = blockIdx.x, job = blockIdx.y, head = blockIdx.z, tid = threadIdx.x;
  const float* red = a.red + (int64_t)head * kWgSlabM;
  const float* dB = red; const float* dM = red + 4096; const float* db = red + kVecOffM;
  const float* dbdyn = a.red + kVecOffM + 64;              // a sum over ALL tokens: every head saw the same dDyn -- take head 0's
  const float* Wq = a.wq + (int64_t)head * 4096; const float* Wk = a.wk + (int64_t)head * 4096; const float* Wv = a.wv + (int64_t)head * 4096;
  const float* Wf = a.fc1_w + head * 64;                  // Wf[n][m] = fc1_w[n * 512 + head * 64 + m]
  float* out = a.out + (int64_t)head * kWgSlab;
  if (job == 0) {          // dW'q[m][b] = sum_a W'k[m][a] dB[a][b]
    mm64_slice(Wk, 64, 1, dB, 64, 1, nullptr, nullptr, out + 0 * 4096, slice, As, Bs);
  } else if (job == 1) {   // dW'k[m][a] = sum_b W'q[m][b] dB[a][b] + cq[m] db[a]
    mm64_slice(Wq, 64, 1, dB, 1, 64, a.cq + head * 64, db, out + 1 * 4096, slice, As, Bs);
  } else if (job == 2) {   // dW'v[m][b] = sum_n Wf[n][m] dM[n][b]
    mm64_slice(Wf, 1, 512, dM, 64, 1, nullptr, nullptr, out + 2 * 4096, slice, As, Bs);
  } else {                 // dWf[n][m] = sum_b dM[n][b] W'v[m][b] + dbdyn[n] cv[m]
    mm64_slice(dM, 64, 1, Wv, 1, 64, dbdyn, a.cv + head * 64, out + 3 * 4096, slice, As, Bs);
  }
  if (slice == 0 && job == 0 && tid < 64) {
    float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
#pragma unroll 8
    for (int x = 0; x < 64; x += 2) {
      s0 += Wk[tid * 64 + x] * db[x]; s1 += Wk[tid * 64 + x + 1] * db[x + 1];
      t0 += Wf[x * 512 + tid] * dbdyn[x]; t1 += Wf[(x + 1) * 512 + tid] * dbdyn[x + 1];
    }
    const float s = s0 + s1, t = t0 + t1;
    out[kVecOff + tid] = s;                          // dcq[m] = sum_a W'k[m][a] db[a]
    out[kVecOff + 64 + tid] = 0.f;                   // dck: the K bias only shifts all scores of a query (no gradient)
    out[kVecOff + 128 + tid] = t;                    // dcv[m] = sum_n Wf[n][m] dbdyn[n]
    out[kVecOff + 192 + tid] = 0.f; out[kVecOff + 256 + tid] = 0.f;     // dK_pad / dV_pad do not exist here
    out[kVecOff + 320 + tid] = head == 0 ? dbdyn[tid] : 0.f;            // dfc1_b = d bdyn
    if (head == 0) {
      float p = 0.f;
      for (int hh = 0; hh < MATCHA_N_HEAD; ++hh) p += a.red[(int64_t)hh * kWgSlabM + kVecOffM + 128 + tid];
      a.dxpad[tid] = p;
    }
  }
}

// ---- slab reduction + un-folding of the LayerNorm affines -------------------------------------------------------
struct UnfoldArgs {
  const float* wslab; int nchunks;
  const float* X; const int32_t* count;
  const float* W[3]; const float* g[3]; const float* b[3];     // original projection weights [512, 64] and LN affines
  float* gW[3]; float* gfc1;                                    // accumulated into
  float* part;                                                  // [32][3][3][64]: {dg, db, dx_hat_pad} partials
};
// grid (4 row slices, 4 matrices {q, k, v, fc1}, 8 heads), 256 threads = 16 rows x 16 lanes (float4)
__global__ __launch_bounds__(256) void fb_unfold_kernel(UnfoldArgs a) {
  __shared__ float red[16][3][64];
  const int slice = blockIdx.x, mat = blockIdx.y, head = blockIdx.z;
  const int tid = threadIdx.x, rl = tid >> 4, c4 = (tid & 15) * 4;
  const int row = slice * 16 + rl;
  const float* base = a.wslab + (int64_t)head * a.nchunks * kWgSlab;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    float4 t0 = s, t1 = s, t2 = s, t3 = s;             // four independent chains: four loads in flight per lane
    const float* pb = base + mat * 4096 + row * 64 + c4;
    int c = 0;
    for (; c + 3 < a.nchunks; c += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(pb + (int64_t)c * kWgSlab), v1 = *reinterpret_cast<const float4*>(pb + (int64_t)(c + 1) * kWgSlab);
      const float4 v2 = *reinterpret_cast<const float4*>(pb + (int64_t)(c + 2) * kWgSlab), v3 = *reinterpret_cast<const float4*>(pb + (int64_t)(c + 3) * kWgSlab);
      t0.x += v0.x; t0.y += v0.y; t0.z += v0.z; t0.w += v0.w;
      t1.x += v1.x; t1.y += v1.y; t1.z += v1.z; t1.w += v1.w;
      t2.x += v2.x; t2.y += v2.y; t2.z += v2.z; t2.w += v2.w;
      t3.x += v3.x; t3.y += v3.y; t3.z += v3.z; t3.w += v3.w;
    }
    for (; c < a.nchunks; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(pb + (int64_t)c * kWgSlab);
      t0.x += v.x; t0.y += v.y; t0.z += v.z; t0.w += v.w;
    }
    s.x = (t0.x + t1.x) + (t2.x + t3.x); s.y = (t0.y + t1.y) + (t2.y + t3.y);
    s.z = (t0.z + t1.z) + (t2.z + t3.z); s.w = (t0.w + t1.w) + (t2.w + t3.w);
  }
  if (mat == 3) {       // dfc1[n][head*64 + k]
    float4* o = reinterpret_cast<float4*>(a.gfc1 + (int64_t)row * 512 + head * 64 + c4);
    float4 v = *o;
    v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
    *o = v;
    return;
  }
  float dc = 0.f, dpad = 0.f;
  for (int c = 0; c < a.nchunks; ++c) {
    const float* vb = base + (int64_t)c * kWgSlab + kVecOff;
    dc += vb[mat * 64 + row];
    if (mat > 0) dpad += vb[192 + (mat - 1) * 64 + row];
  }
  // the padding token's K / V rows are x_hat_pad . W'^T + c: its dK / dV enter dW' and dc like one more token
  const float4 xv = *reinterpret_cast<const float4*>(a.X + (int64_t)a.count[1] * 64 + c4);
  float m, rs;
  ln_row16(xv, m, rs);
  s.x += dpad * (xv.x - m) * rs; s.y += dpad * (xv.y - m) * rs; s.z += dpad * (xv.z - m) * rs; s.w += dpad * (xv.w - m) * rs;
  dc += dpad;
  const int64_t wi = ((int64_t)head * 64 + row) * 64 + c4;
  const float4 W = *reinterpret_cast<const float4*>(a.W[mat] + wi);
  const float4 G = *reinterpret_cast<const float4*>(a.g[mat] + c4), Bv = *reinterpret_cast<const float4*>(a.b[mat] + c4);
  {
    float4* o = reinterpret_cast<float4*>(a.gW[mat] + wi);
    float4 v = *o;
    v.x += s.x * G.x + dc * Bv.x; v.y += s.y * G.y + dc * Bv.y; v.z += s.z * G.z + dc * Bv.z; v.w += s.w * G.w + dc * Bv.w;
    *o = v;
  }
  *reinterpret_cast<float4*>(&red[rl][0][c4]) = make_float4(s.x * W.x, s.y * W.y, s.z * W.z, s.w * W.w);                              // dg
  *reinterpret_cast<float4*>(&red[rl][1][c4]) = make_float4(dc * W.x, dc * W.y, dc * W.z, dc * W.w);                                  // db
  *reinterpret_cast<float4*>(&red[rl][2][c4]) = make_float4(dpad * W.x * G.x, dpad * W.y * G.y, dpad * W.z * G.z, dpad * W.w * G.w);  // dx_hat_pad
  __syncthreads();
  if (tid < 192) {
    const int which = tid >> 6, k = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][which][k];
    a.part[(((int64_t)(head * 4 + slice) * 3 + mat) * 3 + which) * 64 + k] = t;
  }
}

struct Unfold2Args {
  const float* part; const float* wslab; int nchunks;
  float* dg[3]; float* db[3]; float* dfc1_b; float* dxpad;
  int dxpad_add;     // merged heads: dxpad already holds the attention's gradient into the padding token's x_hat (fbm_chain_kernel)
};
// one block, 512 threads: vec 0..5 = {dg, db} x {q, k, v}; vec 6 = dx_hat of the padding token; vec 7 = fc1 bias gradient
__global__ __launch_bounds__(512) void fb_unfold2_kernel(Unfold2Args a) {
  const int vec = threadIdx.x >> 6, k = threadIdx.x & 63;
  float s = 0.f;
  if (vec < 6) {
    const int mat = vec >> 1, which = vec & 1;
    for (int p = 0; p < 32; ++p) s += a.part[(((int64_t)p * 3 + mat) * 3 + which) * 64 + k];
    float* o = (which == 0 ? a.dg[mat] : a.db[mat]) + k;
    *o += s;
  } else if (vec == 6) {
    for (int p = 0; p < 32; ++p) s += a.part[(((int64_t)p * 3 + 1) * 3 + 2) * 64 + k] + a.part[(((int64_t)p * 3 + 2) * 3 + 2) * 64 + k];
    a.dxpad[k] = a.dxpad_add ? a.dxpad[k] + s : s;
  } else {
    for (int c = 0; c < a.nchunks; ++c) s += a.wslab[(int64_t)c * kWgSlab + kVecOff + 320 + k];     // head 0's slabs
    a.dfc1_b[k] += s;
  }
}

// dZ0 = ( LNbwd_noaffine( sum_h dxh[h] ) + dXs ) * (1 - X^2)     (Modules.py:519-521 backward, :270 tanh')
__global__ __launch_bounds__(256) void lnhat_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dxh, int64_t tcap,
                                                        const float* __restrict__ dxpad, const float* __restrict__ dXs,
                                                        float* __restrict__ dZ0, const int32_t* __restrict__ count, int nslab) {
  const int T = count[0];
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c4 = (threadIdx.x & 15) * 4;
  if (t >= T) return;
  const float4 xv = *reinterpret_cast<const float4*>(X + t * 64 + c4);
  float m, rs;
  ln_row16(xv, m, rs);
  const float4 xh = make_float4((xv.x - m) * rs, (xv.y - m) * rs, (xv.z - m) * rs, (xv.w - m) * rs);
  float4 d;
  if (t < T - 1) {
    d = *reinterpret_cast<const float4*>(dxh + t * 64 + c4);
    for (int hd = 1; hd < nslab; ++hd) {
      const float4 v = *reinterpret_cast<const float4*>(dxh + ((int64_t)hd * tcap + t) * 64 + c4);
      d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
    }
  } else {
    d = *reinterpret_cast<const float4*>(dxpad + c4);
  }
  const float a = group_sum<16>((d.x + d.y) + (d.z + d.w)) * (1.f / 64.f);
  const float b = group_sum<16>((d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w)) * (1.f / 64.f);
  const float4 s = *reinterpret_cast<const float4*>(dXs + t * 64 + c4);
  float4 o;
  o.x = (rs * (d.x - a - xh.x * b) + s.x) * (1.f - xv.x * xv.x);
  o.y = (rs * (d.y - a - xh.y * b) + s.y) * (1.f - xv.y * xv.y);
  o.z = (rs * (d.z - a - xh.z * b) + s.z) * (1.f - xv.z * xv.z);
  o.w = (rs * (d.w - a - xh.w * b) + s.w) * (1.f - xv.w * xv.w);
  *reinterpret_cast<float4*>(dZ0 + t * 64 + c4) = o;
}

int chunks_for(int ntiles) {
  const int cus = device_cu_count();
  int c = cus / MATCHA_N_HEAD;
  if (c > kMaxChunks) c = kMaxChunks;
  if (c > ntiles) c = ntiles;
  return c < 1 ? 1 : c;
}

}  // namespace

const float* fused_bwd_dxpad(const float* ws) { return ws + (size_t)MATCHA_N_HEAD * kMaxChunks * kWgSlab + 32 * 3 * 3 * 64; }

size_t fused_bwd_ws_floats(int64_t B, int L) {
  (void)B; (void)L;
  return (size_t)MATCHA_N_HEAD * kMaxChunks * kWgSlab + (size_t)32 * 3 * 3 * 64 + 64;
}

// merged heads: fused_bwdh_kernel -> fbm_chain_kernel -> the LayerNorm un-folding of launch_fused_bwd (one slab per head)
int launch_fused_bwd_merged(const matcha_tensors& p, const float* folded, const float* merged, const float* X, const float* dDyn, const float* dXs,
                            const Ragged& rg, int64_t B, int L, float* dxh, float* ws, matcha_tensors& grads, float* dZ0, hipStream_t st, const float* rimg,
                            bool dx_atomic, bool dx_zeroed, const TailReduceArgs* tail) {
  const int64_t tcap = B * L + 1;
  if (dx_atomic && !dx_zeroed) MATCHA_TRY(zero_async(dxh, (size_t)tcap * 64 * sizeof(float), st));
  int nchunks = 2 * chunks_for(rg.nhalves);                  // two four-wave workgroups per CU
  if (nchunks > kMaxChunks) nchunks = kMaxChunks;
  if (nchunks > rg.nhalves) nchunks = rg.nhalves > 0 ? rg.nhalves : 1;
  float* wslab = ws;                                                         // [8][nchunks][kWgSlabM]
  float* chain = ws + (size_t)MATCHA_N_HEAD * kMaxChunks * kWgSlabM;         // [8][kWgSlab]  (both inside the four-product kernel's slab area)
  float* part = ws + (size_t)MATCHA_N_HEAD * kMaxChunks * kWgSlab;
  float* dxpad = part + 32 * 3 * 3 * 64;
  const size_t wsz = (size_t)MATCHA_N_HEAD * 64 * 64, csz = (size_t)MATCHA_N_HEAD * 64;
  const MergedView mv = merged_view(merged);
  {
    FusedBwdHArgs g;
    g.X = X; g.dDyn = dDyn; g.count = rg.count; g.half_meta = rg.half_meta; g.tok_pos = rg.tok_pos; g.L = L; g.nhalves = rg.nhalves; g.nchunks = nchunks;
    g.mB = mv.B; g.mM = mv.M; g.dxh = dxh; g.tcap = tcap; g.dx_atomic = dx_atomic ? 1 : 0; g.wslab = wslab; g.rimg = rimg;
    const size_t lds = kBwdLdsBytes;
    auto launch = [&](auto kfn) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kfn, dim3(MATCHA_N_HEAD * nchunks), dim3(256), lds, st, g);
    };
    // algorithmic flops: the reference's formulation -- 8 heads x 8 GEMMs of 2*64*64 per token; this kernel EXECUTES half of them
    ProfScope ps(MATCHA_PROF_FUSED_BWD, (double)tcap * MATCHA_N_HEAD * 8.0 * 2.0 * 64.0 * 64.0, st);
    // ROUND-5 FINDING, kept here because it is a property of how this file must be built: compiled with the SLP vectoriser, the ML = 5 instance
    // produced TIMING-DEPENDENT dR rows for hyperedges of exactly 5 nodes when two workgroups shared a CU (~20-120 of 512 workgroups per launch
    // off by 1e-4 of dB_h; never with one workgroup per CU, never for k < 5, never with ML = 6 on the same batch; tools/debug/bwd_slab_diff.py,
    // probes in DESIGN.md 4.1c).  hipcc packs the leftover fifth iteration of the row phase's key loop into v_pk_*_f32 chains whose halves
    // interleave the sig accumulation with the DPP reduction of d_4.  Extra barriers, nops behind the MFMAs and in front of the DPP steps,
    // plain shuffles and unpinned loads did not cure it; -fno-slp-vectorize (no packed f32 in that loop) does -- 0 differing workgroups in
    // every run -- and is faster.  The Makefile sets the flag for the whole library; tests/test_hip_properties.py::test_full_size_train_step_is_reproducible
    // is the run-time guard.
    switch (L <= 2 ? 2 : (L <= 6 ? L : 8)) {
      case 2: launch(fused_bwdh_kernel<2>); break;
      case 3: launch(fused_bwdh_kernel<3>); break;
      case 4: launch(fused_bwdh_kernel<4>); break;
      case 5: launch(fused_bwdh_kernel<5>); break;
      case 6: launch(fused_bwdh_kernel<6>); break;
      default: launch(fused_bwdh_kernel<8>); break;
    }
    MATCHA_CHECK_LAUNCH("fused_bwdh_kernel");
  }
  {
    ChainArgs c;
    c.wslab = wslab; c.nchunks = nchunks; c.red = chain + (size_t)MATCHA_N_HEAD * kWgSlab;
    c.wq = folded; c.wk = folded + wsz; c.wv = folded + 2 * wsz; c.cq = folded + 3 * wsz; c.cv = c.cq + 2 * csz;
    c.fc1_w = p.fc1_w; c.out = chain; c.dxpad = dxpad;
    TailReduceArgs tl;
    if (tail) tl = *tail; else memset(&tl, 0, sizeof(tl));
    const int tail_blocks = tail ? kTailRoleBlocks : 0;
    hipLaunchKernelGGL(fbm_reduce_kernel, dim3((unsigned)(kFbmReduceBlocks + tail_blocks)), dim3(256), 0, st, c, tl, tail_blocks);
    MATCHA_CHECK_LAUNCH("fbm_reduce_kernel");
    hipLaunchKernelGGL(fbm_chain_kernel, dim3(4, 4, MATCHA_N_HEAD), dim3(256), 0, st, c);
    MATCHA_CHECK_LAUNCH("fbm_chain_kernel");
  }
  {
    UnfoldArgs a;
    a.wslab = chain; a.nchunks = 1; a.X = X; a.count = rg.count;
    a.W[0] = p.w_q; a.W[1] = p.w_k; a.W[2] = p.w_v;
    a.g[0] = p.ln_q_g; a.g[1] = p.ln_k_g; a.g[2] = p.ln_v_g;
    a.b[0] = p.ln_q_b; a.b[1] = p.ln_k_b; a.b[2] = p.ln_v_b;
    a.gW[0] = grads.w_q; a.gW[1] = grads.w_k; a.gW[2] = grads.w_v; a.gfc1 = grads.fc1_w;
    a.part = part;
    hipLaunchKernelGGL(fb_unfold_kernel, dim3(4, 4, MATCHA_N_HEAD), dim3(256), 0, st, a);
    MATCHA_CHECK_LAUNCH("fb_unfold_kernel");
    Unfold2Args b;
    b.part = part; b.wslab = chain; b.nchunks = 1;
    b.dg[0] = grads.ln_q_g; b.dg[1] = grads.ln_k_g; b.dg[2] = grads.ln_v_g;
    b.db[0] = grads.ln_q_b; b.db[1] = grads.ln_k_b; b.db[2] = grads.ln_v_b;
    b.dfc1_b = grads.fc1_b; b.dxpad = dxpad; b.dxpad_add = 1;
    hipLaunchKernelGGL(fb_unfold2_kernel, dim3(1), dim3(512), 0, st, b);
    MATCHA_CHECK_LAUNCH("fb_unfold2_kernel");
  }
  if (dZ0) {
    hipLaunchKernelGGL(lnhat_bwd_kernel, dim3((unsigned)cdiv(tcap, 16)), dim3(256), 0, st, X, dxh, tcap, dxpad, dXs, dZ0, rg.count, dx_atomic ? 1 : MATCHA_N_HEAD);
    MATCHA_CHECK_LAUNCH("lnhat_bwd_kernel");
  }
  return MATCHA_OK;
}

}  // namespace matcha
