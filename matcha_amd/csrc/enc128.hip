// Fused attention block for embed_dim 128 (BASELINE configs[3]: hg38 100 kb bins), forward and backward, merged heads in x_hat space:
//
//   x_hat = LayerNorm_noaffine(X);   per head h:   r_i = B'_h x_hat_i + b'_h,   s_ij = r_i . x_hat_j / sqrt(d)  (diagonal masked, the L - k
//   padding keys = the shared padding token's x_hat),   z_i = sum_j p_ij x_hat_j,   dyn_i = sum_h M'_h z_i + c,   Y = dropout(dyn) . non_pad
//
// (Modules.py:519-529 LayerNorms + projections, :417-460 attention, :572 fc1 + dropout, :614 mask).  B'_h, M'_h are the merged matrices of
// model.hip (B_h = W_k[h]^T W_q[h], M_h = Wfc1[:, h] W_v[h]) with the three LayerNorm affines FOLDED in:
//   B'_h = diag(g_k) B_h diag(g_q),  b'_h = g_k * (B_h b_q),  M'_h = M_h diag(g_v),  c = fc1_b + sum_h M_h b_v
// -- the key bias b_k shifts all scores of a query alike and drops out of the softmax; the probabilities sum to one, so the value bias joins c.
// Every head therefore attends the SAME rows, and r / z / dR / dZ ([T, 8 d] tensors, written once and read twice each by the layer-by-layer
// kernels: ~11 GB per 65 536-row step) never exist in HBM.
//
// The products run on the bf16 matrix pipe with three planes per f32 operand (fp32-accurate: six plane products, f32 accumulate; the note is in
// fused_fwd32.hip, the arithmetic in tests/test_cpu_bf16x3.py).  128-wide rows do not fit one wavefront's registers the way the 64-wide rows
// of fused_fwd32.hip do, so BOTH kernels use fused_bwd.hip's scheme: a workgroup of EIGHT wavefronts owns a half tile (<= 31 tokens of whole
// hyperedges, ragged.hip) at a time, every GEMM operand of the token side lives in LDS as bf16 planes, and wavefront w owns 16 of the 128
// output features of every product with its slice of B'_h and M'_h as register fragments:
//
//   forward   workgroup = a chunk of half tiles; OUTER loop over the heads (weights reloaded 8 times per workgroup: 1.5 MB per chunk from L2),
//             inner loop over the chunk's half tiles; the heads' dyn contributions are added into Y in head order by the thread that owns the
//             element (read-modify-write through L2, deterministic), the last head applies bias + dropout
//   backward  workgroup = (head, chunk) like fused_bwdh_kernel; dB'_h / dM'_h accumulate in MFMA accumulators for the whole walk (one slab per
//             workgroup), the heads add their d x_hat into one [T, 128] buffer with float atomics
//
// enc128_unfold_kernel takes the LayerNorm affines back out (gradients of g_q, b_q, g_k, g_v, b_v, fc1_b; dB_h, dM_h for model.hip's merged_chain),
// enc128_lnhat_bwd_kernel is the LayerNorm backward of the summed d x_hat.  Callers that need a bitwise reproducible embedding gradient
// (opts.deterministic, the row-sparse exchange) stay on the layer-by-layer kernels: the atomics order is not fixed.
#include <string.h>

#include "bf16x3.hpp"
#include "kernels.hpp"

namespace matcha {

namespace {

constexpr int kD = 128;
constexpr int kLdF = 132;               // f32 LDS row stride (floats)
constexpr int kPS = 144;                // bf16 per plane row: 72 dwords = 8 mod 64, so that BOTH fragment reads are conflict-free (MI355X guide, LDS): the
                                        // ds_read_b128 row reads (lane groups {0-3, 12-15, 20-27}, ...: banks 8 c16 + 4 kq all distinct) and the
                                        // ds_read_b64_tr_b16 column reads (32-lane halves: rows 4 kq + q of kq = 0, 1 -> banks 8 row + 2 p)
constexpr int kPlane = 32 * kPS;
constexpr int kPT = 3 * kPlane;         // bf16 per three-plane tile (27 648 B)
constexpr float kEps = 1e-5f;
constexpr int kRec = 32 * kD + 256;     // floats per (half tile, head) record: r rows [32][128] + attention probabilities [32][8]
constexpr int kFragHead = 2 * 8 * 4 * 3 * 64;      // u32x4 per head: {B', M'} x 8 waves x 4 steps x 3 planes x 64 lanes (196 608 B)
constexpr int kSlab = 2 * kD * kD + 4 * kD;        // dB'_h dM'_h | db'_h, dxpad, spare, spare
constexpr int kVec = 2 * kD * kD;
constexpr int kMaxChunks = 64;
constexpr int kColBlocks = 1024;

// column fragments (contraction over TOKENS).  The contraction slot <-> token map is free as long as both operands use the same one: slot
// 8 kq + j holds token 4 kq + j (j < 4) or 16 + 4 kq + (j - 4), so that the two 16-lane groups of a 32-lane half read EIGHT CONSECUTIVE rows per
// instruction (conflict-free on 72-dword rows; with tokens 8 kq + j the halves read rows {0-3, 8-11}: 2-way).  The lane's block address: row
// 4 kq + ((lane & 15) >> 2), columns c0 + 4 (lane & 3); lane i of the 16-lane group receives column c0 + i.
typedef Planes<kPS, 16> PL;
__device__ __forceinline__ Frag3 frag_row(const short* __restrict__ p) { return PL::row(p); }
__device__ __forceinline__ void frag_store(short* __restrict__ p, const Frag3& f) { PL::store(p, f); }
__device__ __forceinline__ Frag3 frag_col(const short* __restrict__ p) { return PL::col(p); }

// the same six plane products for TWO accumulators that share one operand, chains interleaved (no MFMA waits on the one before it)
__device__ __forceinline__ void mma6x2_a(f32x4& c0, f32x4& c1, const Frag3& a, const Frag3& b0, const Frag3& b1) {     // shared A operand
  c0 = MFMA16B(a.l, b0.h, c0); c1 = MFMA16B(a.l, b1.h, c1); c0 = MFMA16B(a.h, b0.l, c0); c1 = MFMA16B(a.h, b1.l, c1);
  c0 = MFMA16B(a.m, b0.m, c0); c1 = MFMA16B(a.m, b1.m, c1); c0 = MFMA16B(a.m, b0.h, c0); c1 = MFMA16B(a.m, b1.h, c1);
  c0 = MFMA16B(a.h, b0.m, c0); c1 = MFMA16B(a.h, b1.m, c1); c0 = MFMA16B(a.h, b0.h, c0); c1 = MFMA16B(a.h, b1.h, c1);
}
__device__ __forceinline__ void mma6x2_b(f32x4& c0, f32x4& c1, const Frag3& a0, const Frag3& a1, const Frag3& b) {    // shared B operand
  c0 = MFMA16B(a0.l, b.h, c0); c1 = MFMA16B(a1.l, b.h, c1); c0 = MFMA16B(a0.h, b.l, c0); c1 = MFMA16B(a1.h, b.l, c1);
  c0 = MFMA16B(a0.m, b.m, c0); c1 = MFMA16B(a1.m, b.m, c1); c0 = MFMA16B(a0.m, b.h, c0); c1 = MFMA16B(a1.m, b.h, c1);
  c0 = MFMA16B(a0.h, b.m, c0); c1 = MFMA16B(a1.h, b.m, c1); c0 = MFMA16B(a0.h, b.h, c0); c1 = MFMA16B(a1.h, b.h, c1);
}
// Feature-contraction product of a 32-token tile against this wavefront's weight fragments W[0..3] (the A operand when WEIGHT_A, else the B
// operand): the plane fragments of K step s + 1 are read BEFORE the twelve MFMAs of step s (sched_group_barrier keeps that order: without it
// the compiler issues each read two instructions ahead of its MFMA and every K step waits out the LDS latency)
template <bool WEIGHT_A>
__device__ __forceinline__ void gemm_k128(f32x4& c0, f32x4& c1, const Frag3 (&W)[4], const short* __restrict__ x0) {
  Frag3 p0 = frag_row(x0), p1 = frag_row(x0 + 16 * kPS);
  __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const Frag3 q0 = p0, q1 = p1;
    if (s < 3) { p0 = frag_row(x0 + 32 * (s + 1)); p1 = frag_row(x0 + 16 * kPS + 32 * (s + 1)); }
    if (WEIGHT_A) mma6x2_a(c0, c1, W[s], q0, q1);
    else mma6x2_b(c0, c1, q0, q1, W[s]);
    if (s < 3) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
  }
}

// x_hat of eight features of a 128-wide row spread over 16 lanes (two-pass statistics like row_stats of token_kernels.hip)
__device__ __forceinline__ V8 ln_row8(const V8& x, float msk, float* rstd_out = nullptr, float* mean_out = nullptr) {
  const f2 s2 = (x.a + x.b) + (x.c + x.d);
  const float mean = group_sum16_dpp(s2.x + s2.y) * (1.f / kD);
  const f2 mm = {mean, mean};
  V8 c;
  c.a = x.a - mm; c.b = x.b - mm; c.c = x.c - mm; c.d = x.d - mm;
  f2 q2 = c.a * c.a;
  q2 = __builtin_elementwise_fma(c.b, c.b, q2); q2 = __builtin_elementwise_fma(c.c, c.c, q2); q2 = __builtin_elementwise_fma(c.d, c.d, q2);
  const float q = group_sum16_dpp(q2.x + q2.y);
  const float rs = 1.0f / sqrtf(q * (1.f / kD) + kEps);
  if (rstd_out) *rstd_out = rs;
  if (mean_out) *mean_out = mean;
  return scale8(rs * msk, c);
}

// -DENC_TIMING: per-phase wall-clock (100 MHz) of workgroup 0, printed at the end -- development builds only
#ifdef ENC_TIMING
#define ENC_T(i) do { const long long now__ = wall_clock64(); tph[i] += now__ - tlast; tlast = now__; } while (0)
#else
#define ENC_T(i) do { } while (0)
#endif
#ifndef ENC_ABL
#define ENC_ABL 0                       // in-situ ablations of the forward (wrong results on purpose; tools/debug/abl_fwd32.sh with ABL_SRC=enc128)
#endif
#define ENC_PIN(addr, x) asm volatile("" : "+v"(addr), "+v"((x).a), "+v"((x).b), "+v"((x).c), "+v"((x).d))

// =====================================================================================================================================
// per-step weight forms: folded f32 matrices (backward + un-folding), the forward's register fragments, the bias vectors
// =====================================================================================================================================
struct PrepArgs {
  const float* lwB; const float* lwM;                         // merged matrices of model.hip: B_all [8 d][d], M_all [d][8 d]
  const float* gq; const float* bq; const float* gk; const float* gv; const float* bv; const float* fc1_b;
  float* fold;                                                // [8][2][128][128]: B'_h, M'_h
  u32x4* frag;                                                // [8][kFragHead]
  float* bvec;                                                // [8][128] b'_h, then c [128]
};
__global__ __launch_bounds__(256) void enc128_prep_kernel(PrepArgs a) {
  const int tid = threadIdx.x;
  if (blockIdx.x >= 128) {                                    // bias vectors: 9 blocks, one output per thread
    const int job = blockIdx.x - 128;
    if (tid >= kD) return;
    if (job < 8) {
      const float* row = a.lwB + ((int64_t)job * kD + tid) * kD;
      float s0 = 0.f, s1 = 0.f;
      for (int b = 0; b < kD; b += 2) { s0 += row[b] * a.bq[b]; s1 += row[b + 1] * a.bq[b + 1]; }
      a.bvec[job * kD + tid] = a.gk[tid] * (s0 + s1);
    } else {
      const float* row = a.lwM + (int64_t)tid * 8 * kD;
      float s = 0.f;
      for (int h = 0; h < 8; ++h) {
        float s0 = 0.f, s1 = 0.f;
        for (int b = 0; b < kD; b += 2) { s0 += row[h * kD + b] * a.bv[b]; s1 += row[h * kD + b + 1] * a.bv[b + 1]; }
        s += s0 + s1;
      }
      a.bvec[8 * kD + tid] = a.fc1_b[tid] + s;
    }
    return;
  }
  const int gt = blockIdx.x * 256 + tid;
  const int cg = gt & 15, row = (gt >> 4) & 127, mat = (gt >> 11) & 1, h = gt >> 12;
  float v[8];
  if (mat == 0) {
    const float* src = a.lwB + ((int64_t)h * kD + row) * kD + 8 * cg;
    const float gkr = a.gk[row];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gkr * src[j] * a.gq[8 * cg + j];
  } else {
    const float* src = a.lwM + (int64_t)row * 8 * kD + h * kD + 8 * cg;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[j] * a.gv[8 * cg + j];
  }
  float* fo = a.fold + (((int64_t)h * 2 + mat) * kD + row) * kD + 8 * cg;
  *reinterpret_cast<float4*>(fo) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(fo + 4) = make_float4(v[4], v[5], v[6], v[7]);
  // forward fragment: lane (c16, kq) of wave w, step s holds W[16 w + c16][32 s + 8 kq + {0..7}] -- the A operand of D[feature][token]
  const Frag3 f = split8(v);
  const int wave = row >> 4, c16 = row & 15, s = cg >> 2, kq = cg & 3, lane = kq * 16 + c16;
  u32x4* dst = a.frag + (int64_t)h * kFragHead + ((((int64_t)mat * 8 + wave) * 4 + s) * 3) * 64 + lane;
  dst[0] = f.h; dst[64] = f.m; dst[128] = f.l;
}

// =====================================================================================================================================
// forward
// =====================================================================================================================================
struct FwdArgs {
  const float* X; const int32_t* count; const int32_t* half_meta; const int32_t* tok_pos; const int32_t* tok_slot;
  int L; int nhalves; int nwg;                                // workgroups: each walks an equal share of the half tiles
  const u32x4* frag; const float* bvec;
  float* Y; float* rec;                                       // rec == nullptr: a forward that will not be differentiated
  const uint64_t* seed; float p_drop;
};
constexpr size_t kFwdLdsBytes = (size_t)3 * 32 * kLdF * 4 + (size_t)3 * kPT * 2 + (kD + 64) * 4;

// Two phases per (head, half tile) step, two workgroup barriers:
//   phase Y  attention of step s (vector + LDS work): r rows (Rs) and x_hat rows (Xs[s & 1]) -> z planes (Zp), records
//   phase X  the matrix work around it, back to back: GEMM2 of step s (Zp -> Y), GEMM1 of step s + 1 (Xp[(s + 1) & 1] -> Rs), and the staging of
//            step s + 2 (vector work that runs under the other wavefront's MFMAs); the rows of step s + 3 are fetched for the next phase X
// Steps are the flattened (head, tile) pairs of the workgroup's share, so a head's last GEMM2 and the next head's first GEMM1 share a phase.
template <int ML>
__global__ __launch_bounds__(512) void enc128_fwd_kernel(FwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                                    // [2] x_hat f32: the attention's keys = values
  float* Rs = Xs + 2 * 32 * kLdF;                     // r rows of the step
  short* Xp = reinterpret_cast<short*>(Rs + 32 * kLdF);   // [2] x_hat planes
  short* Zp = Xp + 2 * kPT;                           // z planes
  float* xpad = reinterpret_cast<float*>(Zp + kPT);
  int* tinfo = reinterpret_cast<int*>(xpad + kD);     // [2][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kq = lane >> 4, fb = 16 * wave;
  const int la = tid >> 4, sub = tid & 15;            // staging / attention: 16 lanes per token row, 8 features each
  const int tr = g.count[1];
  int nh = g.count[3];
  if (nh > g.nhalves) nh = g.nhalves;
  const int per = (nh + g.nwg - 1) / g.nwg;
  const int tile_lo = blockIdx.x * per;
  const int tile_hi = tile_lo + per < nh ? tile_lo + per : nh;
  const float inv_temp = 0.08838834764831845f;        // 1 / sqrt(128)

  if (tid < 16) st8(&xpad[8 * tid], ln_row8(ld8(g.X + (int64_t)tr * kD + 8 * tid), 1.f));
  if (blockIdx.x == 0 && tid >= 64 && tid < 96)       // the padding token's Y row (non_pad mask, Modules.py:614)
    *reinterpret_cast<float4*>(g.Y + (int64_t)tr * kD + 4 * (tid - 64)) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tile_lo >= tile_hi) return;

  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (g.p_drop > 0.f) { key = rng_key(*g.seed, kStreamDropFc1); thr = dropout_threshold(g.p_drop); keep_scale = 1.f / (1.f - g.p_drop); }

  const int4* meta = reinterpret_cast<const int4*>(g.half_meta) + tile_lo;
  const int ntile = tile_hi - tile_lo;
  const int nsteps = MATCHA_N_HEAD * ntile;
  // staging registers: this thread's 8 floats of its row + the row's position word
  V8 xn;
  int tpn;
#define ENC_GLOAD(M)                                                                                     \
  do {                                                                                                   \
    const int64_t tok__ = (M).x + (la < (M).y ? la : ((M).y > 0 ? (M).y - 1 : 0));                        \
    xn = ld8(g.X + tok__ * kD + 8 * sub);                                                                \
    tpn = g.tok_pos[tok__];                                                                              \
  } while (0)
#define ENC_STAGE(BUF, NREAL)                                                                            \
  do {                                                                                                   \
    const V8 xh__ = ln_row8(xn, la < (NREAL) ? 1.f : 0.f);                                               \
    st8(&Xs[(BUF) * 32 * kLdF + la * kLdF + 8 * sub], xh__);                                             \
    frag_store(Xp + (BUF) * kPT + la * kPS + 8 * sub, split8(xh__));                                     \
    if (sub == 0) tinfo[(BUF) * 32 + la] = la < (NREAL) ? ((la - (tpn & 255)) | (tpn & ~255)) : 0;       \
  } while (0)
  // this wavefront's 16 output features of B'_h (GEMM1) / M'_h (GEMM2) as register fragments
  Frag3 Bw[4], Mw[4];
  f32x4 bias4, cvec4 = {0.f, 0.f, 0.f, 0.f};
#define ENC_LOAD_B(H)                                                                                    \
  do {                                                                                                   \
    const u32x4* fp__ = g.frag + (int64_t)(H) * kFragHead + (int64_t)wave * (4 * 3 * 64) + lane;         \
    _Pragma("unroll") for (int s__ = 0; s__ < 4; ++s__) {                                                \
      Bw[s__].h = fp__[(s__ * 3 + 0) * 64]; Bw[s__].m = fp__[(s__ * 3 + 1) * 64]; Bw[s__].l = fp__[(s__ * 3 + 2) * 64]; \
    }                                                                                                    \
    bias4 = *reinterpret_cast<const f32x4*>(g.bvec + (H) * kD + fb + 4 * kq);                            \
  } while (0)
#define ENC_LOAD_M(H)                                                                                    \
  do {                                                                                                   \
    const u32x4* fp__ = g.frag + (int64_t)(H) * kFragHead + (int64_t)(8 + wave) * (4 * 3 * 64) + lane;   \
    _Pragma("unroll") for (int s__ = 0; s__ < 4; ++s__) {                                                \
      Mw[s__].h = fp__[(s__ * 3 + 0) * 64]; Mw[s__].m = fp__[(s__ * 3 + 1) * 64]; Mw[s__].l = fp__[(s__ * 3 + 2) * 64]; \
    }                                                                                                    \
    if ((H) == MATCHA_N_HEAD - 1) cvec4 = *reinterpret_cast<const f32x4*>(g.bvec + 8 * kD + fb + 4 * kq); \
  } while (0)
  // r^T = B'_h x_hat^T + b'_h: lane (c16, kq) ends with token c16 (+ 16), features fb + 4 kq + {0..3}
#define ENC_GEMM1(BUF)                                                                                   \
  do {                                                                                                   \
    f32x4 acc0__ = bias4, acc1__ = bias4;                                                                \
    if (!(ENC_ABL & 2)) gemm_k128<true>(acc0__, acc1__, Bw, Xp + (BUF) * kPT + c16 * kPS + 8 * kq);      \
    *reinterpret_cast<f32x4*>(&Rs[c16 * kLdF + fb + 4 * kq]) = acc0__;                                   \
    *reinterpret_cast<f32x4*>(&Rs[(16 + c16) * kLdF + fb + 4 * kq]) = acc1__;                            \
  } while (0)

  // ---- prologue: stage steps 0 and 1, GEMM1 of step 0, fetch the rows of step 2 ----
  int4 mA = meta[0];                                  // step s: tile s % ntile, head s / ntile
  int4 mB = meta[1 < ntile ? 1 : 0];
  int4 mC = meta[2 % ntile];
  int4 mD = mC;
  ENC_GLOAD(mA);
  ENC_LOAD_B(0);
  ENC_STAGE(0, mA.y);
  if (nsteps > 1) ENC_GLOAD(mB);
  __syncthreads();
  ENC_GEMM1(0);
  if (nsteps > 1) ENC_STAGE(1, mB.y);
  if (nsteps > 2) ENC_GLOAD(mC);
  __syncthreads();
#ifdef ENC_TIMING
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = wall_clock64();
#endif
  int tiA = 0, hA = 0;                                // tile index (inside the share) and head of step st
  int ti3 = 3 % ntile;                                // tile index of step st + 3
  for (int st = 0; st < nsteps; ++st) {
    const int t0 = mA.x, n_real = mA.y, buf = st & 1;
    const int tiB = tiA + 1 == ntile ? 0 : tiA + 1;
    const int hB = tiB == 0 ? hA + 1 : hA;
    // ---- phase Y: attention in x_hat space, 16 lanes per query token ----
    {
      V8 z = zero8();
      const float* Xb = Xs + buf * 32 * kLdF;
      if (!(ENC_ABL & 1) && la < n_real) {
        const int ia = tinfo[buf * 32 + la];
        const int li0 = ia & 255, k = ia >> 8, pos = la - li0, n_pad = g.L - k;
        const float padf = (float)n_pad;
        const V8 r = ld8(&Rs[la * kLdF + 8 * sub]);
        const V8 xp8 = ld8(&xpad[8 * sub]);
        V8 xk[ML];
        float s[ML];
#pragma unroll
        for (int j = 0; j < ML; ++j) xk[j] = ld8(&Xb[(li0 + (j < k ? j : 0)) * kLdF + 8 * sub]);
        float mx = -3.4e38f;
#pragma unroll
        for (int j = 0; j < ML; ++j) {
          float v = group_sum16_dpp(dot8(r, xk[j])) * inv_temp;
          if (j == pos) v = -1e32f;
          s[j] = v;
          if (j < k) mx = fmaxf(mx, v);
        }
        float sp = group_sum16_dpp(dot8(r, xp8)) * inv_temp;
        if (n_pad > 0) mx = fmaxf(mx, sp);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < ML; ++j) { s[j] = (j < k) ? __expf(s[j] - mx) : 0.f; den += s[j]; }
        sp = (n_pad > 0) ? __expf(sp - mx) : 0.f;
        den += padf * sp;
        const float inv = 1.f / den;
        sp *= inv;
        z = scale8(padf * sp, xp8);
#pragma unroll
        for (int j = 0; j < ML; ++j) { s[j] *= inv; axpy8(z, s[j], xk[j]); }
        if (!(ENC_ABL & 32) && g.rec) {
          float* rb = g.rec + ((int64_t)(tile_lo + tiA) * MATCHA_N_HEAD + hA) * kRec;
          float* rr = rb + la * kD + 8 * sub;
          __builtin_nontemporal_store((f32x4){r.a.x, r.a.y, r.b.x, r.b.y}, reinterpret_cast<f32x4*>(rr));
          __builtin_nontemporal_store((f32x4){r.c.x, r.c.y, r.d.x, r.d.y}, reinterpret_cast<f32x4*>(rr) + 1);
          if (sub == 0) {
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = j < ML ? s[j < ML ? j : 0] : 0.f;
            if (n_pad > 0) w[7] = sp;
            float* pr = rb + 32 * kD + la * 8;
            *reinterpret_cast<float4*>(pr) = make_float4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<float4*>(pr + 4) = make_float4(w[4], w[5], w[6], w[7]);
          }
        }
      }
      frag_store(Zp + la * kPS + 8 * sub, split8(z));
    }
    ENC_T(2);
    __syncthreads();
    ENC_T(7);
    // ---- phase X ----
    if (tiA == 0) ENC_LOAD_M(hA);
    // dyn^T (this head's share) = M'_h z^T, added into Y by the owner of the element; the last head adds c and applies the dropout
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, old0 = acc0, old1 = acc0;
      float* y0 = g.Y + ((int64_t)t0 + c16) * kD + fb + 4 * kq;
      float* y1 = y0 + 16 * kD;
      const bool ok0 = c16 < n_real, ok1 = 16 + c16 < n_real;
      if (!(ENC_ABL & 16) && hA > 0) {                // the earlier heads' sum: in flight under the MFMAs, added behind them
        if (ok0) old0 = *reinterpret_cast<const f32x4*>(y0);
        if (ok1) old1 = *reinterpret_cast<const f32x4*>(y1);
      }
      if (!(ENC_ABL & 4)) gemm_k128<true>(acc0, acc1, Mw, Zp + c16 * kPS + 8 * kq);
      acc0 += old0; acc1 += old1;
      if (hA == MATCHA_N_HEAD - 1) {
        acc0 += cvec4; acc1 += cvec4;
        if (g.p_drop > 0.f) {
          const uint32_t col = (uint32_t)(fb + 4 * kq);
          if (ok0) {
            const uint32_t crow = (uint32_t)g.tok_slot[t0 + c16];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[e] = (rng_u32(key, crow, col + e) >= thr) ? acc0[e] * keep_scale : 0.f;
          }
          if (ok1) {
            const uint32_t crow = (uint32_t)g.tok_slot[t0 + 16 + c16];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[e] = (rng_u32(key, crow, col + e) >= thr) ? acc1[e] * keep_scale : 0.f;
          }
        }
      }
      if ((!(ENC_ABL & 16) || hA == 7) && ok0) *reinterpret_cast<f32x4*>(y0) = acc0;
      if ((!(ENC_ABL & 16) || hA == 7) && ok1) *reinterpret_cast<f32x4*>(y1) = acc1;
    }
    ENC_T(3);
    if (st + 1 < nsteps) {
      if (tiB == 0) ENC_LOAD_B(hB);
      ENC_GEMM1(buf ^ 1);
    }
    ENC_T(1);
    if (!(ENC_ABL & 8) && st + 2 < nsteps) ENC_STAGE(buf, mC.y);         // step st + 2 takes over the buffers step st is done with
    if (st + 3 < nsteps) { mD = meta[ti3]; ENC_GLOAD(mD); }
    ENC_T(4);
    __syncthreads();
    ENC_T(7);
    mA = mB; mB = mC; mC = mD;
    tiA = tiB; hA = hB;
    ti3 = ti3 + 1 == ntile ? 0 : ti3 + 1;
  }
#ifdef ENC_TIMING
  if (blockIdx.x == 0 && (tid == 0 || tid == 448))
    printf("enc128_fwd wg0 wave %d us: gemm1 %.1f attn %.1f gemm2 %.1f stage %.1f barrier-wait %.1f (tiles %d x 8 heads)\n", tid >> 6,
           tph[1] * 0.01, tph[2] * 0.01, tph[3] * 0.01, tph[4] * 0.01, tph[7] * 0.01, ntile);
#endif
#undef ENC_GLOAD
#undef ENC_STAGE
#undef ENC_LOAD_B
#undef ENC_LOAD_M
#undef ENC_GEMM1
}

// =====================================================================================================================================
// backward
// =====================================================================================================================================
// Row phase of token i (16 lanes, 8 features each): keys = values = the x_hat rows, ONE pass (the algebra is fused_bwd.hip's attn_row8_kv):
//   z_i = sum_j w_j x_j,  d_j = dz_i . x_j,  sig = sum_j w_j d_j,  A = sum_j (w_j d_j) x_j,  d r_i = (A - sig z_i) / temp,  dS_ij = w_j (d_j - sig) / temp
template <int ML>
__device__ __forceinline__ void attn_row16(const float* __restrict__ Rs, const float* __restrict__ Xs, const float* __restrict__ Fs,
                                           const float* __restrict__ xpad, const float* __restrict__ Ps, float* __restrict__ dSs, int li, int li0,
                                           int k, int n_pad, int sub, float inv_temp, V8& o, V8& gq, V8& accP) {
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
  const int ro0 = li0 * kLdF + 8 * sub;
  const V8 go = ld8(&Fs[li * kLdF + 8 * sub]);
  float dsp_raw;
  V8 a;
  {
    const V8 vp = ld8(xpad + 8 * sub);
    dsp_raw = group_sum16_dpp(dot8(go, vp));
    o = scale8(ppf, vp);
    a = scale8(ppf * dsp_raw, vp);
  }
  float sig = ppf * dsp_raw;
  V8 vn = ld8(&Xs[ro0]);
#pragma unroll
  for (int j = 0; j < ML; ++j) {
    const V8 v = vn;
    if (j + 1 < ML) {
      int ad = ro0 + (j + 1 < k ? j + 1 : 0) * kLdF;
      ENC_PIN(ad, o);
      vn = ld8(&Xs[ad]);
    }
    axpy8(o, p[j], v);
    const float d = group_sum16_dpp(dot8(go, v));
    ds[j] = d;
    const float wd = p[j] * d;
    sig += wd;
    axpy8(a, wd, v);
  }
#pragma unroll
  for (int j = 0; j < ML; ++j) ds[j] = p[j] * (ds[j] - sig) * inv_temp;
  const float dspf = padf * (pp * (dsp_raw - sig) * inv_temp);
  axpy8(accP, ppf, go);                                    // the padding token as a value: sum_i n_pad p_i,pad dz_i
  {
    const float ns = -sig;
    axpy8(a, ns, o);
    gq = scale8(inv_temp, a);
  }
  {
    int ad = li * kLdF + 8 * sub;
    ENC_PIN(ad, gq);
    const V8 q = ld8(&Rs[ad]);
    axpy8(accP, dspf, q);                                  // ... and as a key: sum_i n_pad dS_i,pad r_i
  }
  if (sub == 1) {
    float* dst = dSs + li * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(ds[0], ds[1 % ML], ML > 2 ? ds[2 % ML] : 0.f, ML > 3 ? ds[3 % ML] : 0.f);
    if (ML > 4) *reinterpret_cast<float4*>(dst + 4) = make_float4(ds[4 % ML], ML > 5 ? ds[5 % ML] : 0.f, ML > 6 ? ds[6 % ML] : 0.f, ML > 7 ? ds[7 % ML] : 0.f);
  }
}
// Column phase of token j: d x_hat_j (attention part) = sum_i dS_ij r_i + p_ij dz_i over the hyperedge's rows
template <int ML>
__device__ __forceinline__ V8 attn_col16(const float* __restrict__ Rs, const float* __restrict__ Fs, const float* __restrict__ Ps,
                                         const float* __restrict__ dSs, int li, int li0, int k, int sub) {
  const int jj = li - li0;
  V8 gk = zero8();
  const int ro0 = li0 * kLdF + 8 * sub;
  V8 qn = ld8(&Rs[ro0]), gn = ld8(&Fs[ro0]);
#pragma unroll
  for (int i = 0; i < ML; ++i) {
    const V8 q = qn, go = gn;
    const int ri = li0 + (i < k ? i : 0);
    const float pl = Ps[ri * 8 + jj], dl = dSs[ri * 8 + jj];
    const float pij = (i < k) ? pl : 0.f, dsij = (i < k) ? dl : 0.f;
    if (i + 1 < ML) {
      int a = ro0 + (i + 1 < k ? i + 1 : 0) * kLdF;
      ENC_PIN(a, gk);
      qn = ld8(&Rs[a]); gn = ld8(&Fs[a]);
    }
    axpy8(gk, pij, go);
    axpy8(gk, dsij, q);
  }
  return gk;
}

struct BwdArgs {
  const float* X; const float* dDyn; const int32_t* count; const int32_t* half_meta; const int32_t* tok_pos;
  int L; int nhalves; int nchunks;
  const float* fold;                                // [8][2][128][128]
  float* dxh;                                       // [tcap][128], zeroed by the launcher: every head adds with float atomics
  float* wslab;                                     // [8][nchunks][kSlab]
  const float* rec;
};
constexpr size_t kBwdLdsBytes = (size_t)2 * 32 * kLdF * 4 + (size_t)4 * kPT * 2 + (kD + 256 + 256 + 32) * 4;

template <int ML>
__global__ __launch_bounds__(512) void enc128_bwd_kernel(BwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                                   // x_hat f32
  float* Gs = lds + 32 * kLdF;                       // the attention's gradient into the x_hat rows
  short* Xp = reinterpret_cast<short*>(lds + 2 * 32 * kLdF);
  short* Dp = Xp + kPT;                              // dDyn planes
  short* RBp = Dp + kPT;                             // r f32 -> dR planes
  short* FBp = RBp + kPT;                            // dZ f32 -> Z planes
  float* Rs = reinterpret_cast<float*>(RBp);
  float* Fs = reinterpret_cast<float*>(FBp);
  float* xpad = reinterpret_cast<float*>(FBp + kPT);
  float* dSs = xpad + kD;              // [32][8]
  float* Ps = dSs + 256;               // [32][8]
  int* tinfo = reinterpret_cast<int*>(Ps + 256);

  const int tid = threadIdx.x;
  int head, chunk;
  if ((g.nchunks & 7) == 0) {          // the eight heads of a chunk on one XCD: their d x_hat atomics meet in that L2
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
  const float inv_temp = 0.08838834764831845f;

  if (tid < 16) st8(&xpad[8 * tid], ln_row8(ld8(g.X + (int64_t)tr * kD + 8 * tid), 1.f));

  // weight-gradient accumulators: rows 16 i + 4 kq + reg, column fb + c16 of dB'_h and dM'_h
  f32x4 ab[8], am[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ab[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; am[i] = ab[i]; }
  V8 accP = zero8();                   // d x_hat of the padding token (as key and as value)
  V8 accR = zero8();                   // column sums of dR = db'_h

  const int4* meta = reinterpret_cast<const int4*>(g.half_meta);
  const int4 mzero = make_int4(0, 0, 0, 0);
  int4 mc = tile_lo < tile_hi ? meta[tile_lo] : mzero;
  int4 mn = tile_lo + 1 < tile_hi ? meta[tile_lo + 1] : mzero;
  V8 xn, dn, rn;
  f32x4 pn = {0.f, 0.f, 0.f, 0.f};
  int tpn = 0;
#define ENCB_ROWS_GLOAD(M)                                                                               \
  do {                                                                                                   \
    const int la__ = tid >> 4, sub__ = tid & 15;                                                         \
    const int64_t tok__ = (M).x + (la__ < (M).y ? la__ : ((M).y > 0 ? (M).y - 1 : 0));                    \
    xn = ld8(g.X + tok__ * kD + 8 * sub__);                                                              \
    dn = ld8(g.dDyn + tok__ * kD + 8 * sub__);                                                           \
    tpn = g.tok_pos[tok__];                                                                              \
  } while (0)
#define ENCB_REC_GLOAD(HALF)                                                                             \
  do {                                                                                                   \
    const float* r__ = g.rec + ((int64_t)(HALF) * MATCHA_N_HEAD + head) * kRec;                          \
    const f32x4* q__ = reinterpret_cast<const f32x4*>(r__ + (tid >> 4) * kD + 8 * (tid & 15));           \
    const f32x4 a__ = __builtin_nontemporal_load(q__), b__ = __builtin_nontemporal_load(q__ + 1);        \
    rn.a = f2{a__[0], a__[1]}; rn.b = f2{a__[2], a__[3]}; rn.c = f2{b__[0], b__[1]}; rn.d = f2{b__[2], b__[3]}; \
    if (tid < 64) pn = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r__ + 32 * kD) + tid);  \
  } while (0)
  // M'_h and B'_h as register fragments for the whole walk: lane (c16, kq), step s holds W[32 s + 8 kq + {0..7}][fb + c16] -- the A operand of
  // dZ^T = M'^T dDyn^T (rows = features of dZ) and the B operand of d x_hat = dR B' (columns = features)
  Frag3 Mf[4], Bf[4];
  {
    const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kq = lane >> 4, fb = 16 * wave;
    ENCB_ROWS_GLOAD(mc);
    if (tile_lo < tile_hi) ENCB_REC_GLOAD(tile_lo);
    const float* bp = g.fold + ((int64_t)head * 2 + 0) * kD * kD + (8 * kq) * kD + fb + c16;
    const float* mp = g.fold + ((int64_t)head * 2 + 1) * kD * kD + (8 * kq) * kD + fb + c16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float vm[8], vb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { vm[j] = mp[(32 * s + j) * kD]; vb[j] = bp[(32 * s + j) * kD]; }
      Mf[s] = split8(vm); Bf[s] = split8(vb);
    }
  }

#ifdef ENC_TIMING
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = wall_clock64();
#endif
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int4 mnn = tile + 2 < tile_hi ? meta[tile + 2] : mzero;
    const int t0 = mc.x, n_real = mc.y;
    __syncthreads();                                  // the previous half tile's GEMMs are done with every tile
    ENC_T(7);
    // per-lane indices re-derived from an opaque copy of the thread id (loop-invariant addresses are hoisted and spilled otherwise)
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int c16 = lane & 15, kq = lane >> 4;
    const int fb = 16 * wave;
    const int la = tid_ >> 4, sub = tid_ & 15;
    // ---- stage this half tile (its rows were fetched during the previous one's GEMMs) ----
    {
      const float msk = la < n_real ? 1.f : 0.f;
      const V8 xh = ln_row8(xn, msk);
      st8(&Xs[la * kLdF + 8 * sub], xh);
      frag_store(Xp + la * kPS + 8 * sub, split8(xh));
      frag_store(Dp + la * kPS + 8 * sub, split8(scale8(msk, dn)));
      st8(&Rs[la * kLdF + 8 * sub], rn);
      if (sub == 0) tinfo[la] = la < n_real ? ((la - (tpn & 255)) | (tpn & ~255)) : 0;
      if (tid_ < 64) reinterpret_cast<f32x4*>(Ps)[tid_] = pn;
    }
    ENC_T(0);
    __syncthreads();
    ENC_T(7);
    // ---- dZ^T = M'^T . dDyn^T: lane (c16, kq) ends with token c16 (+ 16) and features fb + 4 kq + {0..3} ----
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const short* dp = Dp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const Frag3 b0 = frag_row(dp + 32 * s), b1 = frag_row(dp + 16 * kPS + 32 * s);
        acc0 = mma6(acc0, Mf[s], b0); acc1 = mma6(acc1, Mf[s], b1);
      }
      *reinterpret_cast<f32x4*>(&Fs[c16 * kLdF + fb + 4 * kq]) = acc0;
      *reinterpret_cast<f32x4*>(&Fs[(16 + c16) * kLdF + fb + 4 * kq]) = acc1;
    }
    ENC_T(1);
    __syncthreads();
    ENC_T(7);
    // ---- attention forward + backward in x_hat space: 16 lanes per token, all 32 rows in one pass ----
    {
      V8 o0 = zero8(), q0 = zero8();
      const bool acta = la < n_real;
      int ia = 0;
      if (acta) { ia = tinfo[la]; attn_row16<ML>(Rs, Xs, Fs, xpad, Ps, dSs, la, ia & 255, ia >> 8, g.L - (ia >> 8), sub, inv_temp, o0, q0, accP); }
      __builtin_amdgcn_sched_barrier(0);
      ENC_T(2);
      __syncthreads();
      ENC_T(7);
      if (acta) st8(&Gs[la * kLdF + 8 * sub], attn_col16<ML>(Rs, Fs, Ps, dSs, la, ia & 255, ia >> 8, sub));
      else st8(&Gs[la * kLdF + 8 * sub], zero8());
      ENC_T(3);
      __syncthreads();                                // every column phase is done with the r and dZ rows: they become the dR and Z PLANES
      frag_store(FBp + la * kPS + 8 * sub, split8(o0));
      frag_store(RBp + la * kPS + 8 * sub, split8(q0));
      add8(accR, q0);
    }
    ENC_T(4);
    __syncthreads();
    ENC_T(7);
    // ---- this head's share of d x_hat = dR B' + Gs: rows = tokens 4 kq + reg (+ 16), columns = features fb + c16 (one atomic = 4 rows x 64 B) ----
    {
      const float* gp = Gs + (4 * kq) * kLdF + fb + c16;
      f32x4 dx0 = {gp[0], gp[kLdF], gp[2 * kLdF], gp[3 * kLdF]};
      f32x4 dx1 = {gp[16 * kLdF], gp[17 * kLdF], gp[18 * kLdF], gp[19 * kLdF]};
      const short* arow = RBp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const Frag3 a0 = frag_row(arow + 32 * s), a1 = frag_row(arow + 16 * kPS + 32 * s);
        dx0 = mma6(dx0, a0, Bf[s]); dx1 = mma6(dx1, a1, Bf[s]);
      }
      float* out = g.dxh + ((int64_t)t0 + 4 * kq) * kD + fb + c16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        if (4 * kq + reg < n_real) unsafeAtomicAdd(out + reg * kD, dx0[reg]);
        if (16 + 4 * kq + reg < n_real) unsafeAtomicAdd(out + (16 + reg) * kD, dx1[reg]);
      }
    }
    ENC_T(5);
    if (tile + 1 < tile_hi) ENCB_REC_GLOAD(tile + 1);   // next half tile's r rows and probabilities: in flight during the weight-gradient GEMMs
    // ---- weight gradients: dB'[a][b] += sum_t dR[t][a] x_hat[t][b];  dM'[n][b] += sum_t dDyn[t][n] Z[t][b]: ONE 32-token step, column fragments ----
    {
      const int blk = ((4 * kq + ((lane & 15) >> 2)) * kPS) + 4 * (lane & 3);
      const Frag3 xb = frag_col(Xp + blk + fb), zb = frag_col(FBp + blk + fb);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const Frag3 ri = frag_col(RBp + blk + 16 * i);
        ab[i] = mma6(ab[i], ri, xb);
      }
      ENCB_ROWS_GLOAD(mn);                            // next half tile's rows: in flight during the second weight-gradient product and the barrier
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const Frag3 di = frag_col(Dp + blk + 16 * i);
        am[i] = mma6(am[i], di, zb);
      }
    }
    ENC_T(6);
    mc = mn; mn = mnn;
  }
#ifdef ENC_TIMING
  if (blockIdx.x == 0 && (tid == 0 || tid == 448))
    printf("enc128_bwd wg0 wave %d us: stage %.1f dZ %.1f attn-row %.1f attn-col %.1f attn-write %.1f dx %.1f tn %.1f barrier-wait %.1f (halves %d)\n", tid >> 6,
           tph[0] * 0.01, tph[1] * 0.01, tph[2] * 0.01, tph[3] * 0.01, tph[4] * 0.01, tph[5] * 0.01, tph[6] * 0.01, tph[7] * 0.01, tile_hi - tile_lo);
#endif
#undef ENCB_ROWS_GLOAD
#undef ENCB_REC_GLOAD

  // ---- workgroup slab ----
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kq = lane >> 4, fb = 16 * wave;
  float* slab = g.wslab + ((int64_t)head * g.nchunks + chunk) * kSlab;
  {
    const int col = fb + c16;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = 16 * i + 4 * kq + reg;
        slab[row * kD + col] = ab[i][reg];
        slab[kD * kD + row * kD + col] = am[i][reg];
      }
  }
  // column sums of dR and d x_hat of the padding token: the 4 lane groups with equal `sub` of a wave (fixed xor tree), then the 8 waves in order
  float* redr = lds;                    // [8][128]
  float* redp = lds + 8 * kD;           // [8][128]
  {
    const int sub = tid & 15;
    const float accv[16] = {accP.a.x, accP.a.y, accP.b.x, accP.b.y, accP.c.x, accP.c.y, accP.d.x, accP.d.y,
                            accR.a.x, accR.a.y, accR.b.x, accR.b.y, accR.c.x, accR.c.y, accR.d.x, accR.d.y};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = accv[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) (i < 8 ? redp : redr)[wave * kD + 8 * sub + (i & 7)] = v;
    }
  }
  __syncthreads();
  if (tid < kD) {
    float r = 0.f, p = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { r += redr[w * kD + tid]; p += redp[w * kD + tid]; }
    slab[kVec + tid] = r;               // db'_h partial
    slab[kVec + kD + tid] = p;          // dxpad partial
  }
}

// chunk sums of every slab element in chunk order (grid.y = head); grid.y == 8: the column sums of dDyn from their block partials
struct ReduceArgs { const float* wslab; int nchunks; float* red; const float* colpart; float* dc; };
__global__ __launch_bounds__(256) void enc128_reduce_kernel(ReduceArgs a) {
  const int head = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (head == MATCHA_N_HEAD) {
    if (i >= kD) return;
    float s = 0.f;
    for (int b = 0; b < kColBlocks; ++b) s += a.colpart[b * kD + i];
    a.dc[i] = s;
    return;
  }
  if (i >= kSlab) return;
  const float* base = a.wslab + (int64_t)head * a.nchunks * kSlab + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int c = 0;
  for (; c + 7 < a.nchunks; c += 8) {
    s0 += base[(int64_t)c * kSlab]; s1 += base[(int64_t)(c + 1) * kSlab]; s2 += base[(int64_t)(c + 2) * kSlab]; s3 += base[(int64_t)(c + 3) * kSlab];
    s4 += base[(int64_t)(c + 4) * kSlab]; s5 += base[(int64_t)(c + 5) * kSlab]; s6 += base[(int64_t)(c + 6) * kSlab]; s7 += base[(int64_t)(c + 7) * kSlab];
  }
  for (; c < a.nchunks; ++c) s0 += base[(int64_t)c * kSlab];
  a.red[(int64_t)head * kSlab + i] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}

// column sums of dDyn over the valid token rows: colpart[block][128] (the fc1 bias gradient and the value-bias term of dM_h)
__global__ __launch_bounds__(256) void enc128_colsum_kernel(const float* __restrict__ dDyn, const int32_t* __restrict__ count, float* __restrict__ colpart) {
  __shared__ float red[8][kD];
  const int T = count[0];
  const int c4 = (threadIdx.x & 31) * 4, rl = threadIdx.x >> 5;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t t = (int64_t)blockIdx.x * 8 + rl; t < T; t += (int64_t)gridDim.x * 8) {
    const float4 v = *reinterpret_cast<const float4*>(dDyn + t * kD + c4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(&red[rl][c4]) = s;
  __syncthreads();
  if (threadIdx.x < kD) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
    colpart[blockIdx.x * kD + threadIdx.x] = t;
  }
}

// The LayerNorm affines back out of (dB'_h, db'_h, dM'_h, dc), block i = feature index i:
//   dB_h[a][b] = g_k[a] g_q[b] dB'[a][b] + g_k[a] db'[a] b_q[b]            dM_h[n][b] = dM'[n][b] g_v[b] + dc[n] b_v[b]
//   dg_k[a] = sum_h,b dB'[a][b] B[a][b] g_q[b] + db'[a] sum_b B[a][b] b_q[b]   dg_q[b] = sum_h,a dB'[a][b] g_k[a] B[a][b]
//   db_q[b] = sum_h,a g_k[a] db'[a] B[a][b]      dg_v[b] = sum_h,n dM'[n][b] M[n][b]      db_v[b] = sum_h,n dc[n] M[n][b]      dfc1_b = dc
// (b_k has no gradient: it shifts all scores of a query alike.)  The padding token's d x_hat (summed over heads) is added into its dxh row.
struct UnfoldArgs {
  const float* red; const float* dc; const float* lwB; const float* lwM;
  const float* gq; const float* bq; const float* gk; const float* gv; const float* bv;
  float* lwdB; float* lwdM;
  float* dgq; float* dbq; float* dgk; float* dgv; float* dbv; float* dfc1_b;
  float* dxh; const int32_t* count;
};
__global__ __launch_bounds__(256) void enc128_unfold_kernel(UnfoldArgs a) {
  __shared__ float red[5][256];
  const int i = blockIdx.x, tid = threadIdx.x;
  float s_gk = 0.f, s_gq = 0.f, s_bq = 0.f, s_gv = 0.f, s_bv = 0.f;
  const float gki = a.gk[i], dci = a.dc[i];
  for (int e = tid; e < 8 * kD; e += 256) {
    const int h = e >> 7, b = e & 127;
    const float* rh = a.red + (int64_t)h * kSlab;
    // row i of head h
    {
      const float dBp = rh[i * kD + b], Bv = a.lwB[((int64_t)h * kD + i) * kD + b], dbp = rh[kVec + i];
      a.lwdB[((int64_t)h * kD + i) * kD + b] = gki * (a.gq[b] * dBp + dbp * a.bq[b]);
      s_gk += Bv * (dBp * a.gq[b] + dbp * a.bq[b]);
      const float dMp = rh[kD * kD + i * kD + b];
      a.lwdM[(int64_t)i * 8 * kD + h * kD + b] = dMp * a.gv[b] + dci * a.bv[b];
    }
    // column i of head h (row index = b here)
    {
      const float dBp = rh[b * kD + i], Bv = a.lwB[((int64_t)h * kD + b) * kD + i], gkb = a.gk[b];
      s_gq += dBp * gkb * Bv;
      s_bq += gkb * rh[kVec + b] * Bv;
      const float dMp = rh[kD * kD + b * kD + i], Mv = a.lwM[(int64_t)b * 8 * kD + h * kD + i];
      s_gv += dMp * Mv;
      s_bv += a.dc[b] * Mv;
    }
  }
  red[0][tid] = s_gk; red[1][tid] = s_gq; red[2][tid] = s_bq; red[3][tid] = s_gv; red[4][tid] = s_bv;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int v = 0; v < 5; ++v) red[v][tid] += red[v][tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.dgk[i] += red[0][0]; a.dgq[i] += red[1][0]; a.dbq[i] += red[2][0]; a.dgv[i] += red[3][0]; a.dbv[i] += red[4][0];
    a.dfc1_b[i] += dci;
    float p = 0.f;
    for (int h = 0; h < MATCHA_N_HEAD; ++h) p += a.red[(int64_t)h * kSlab + kVec + kD + i];
    a.dxh[(int64_t)a.count[1] * kD + i] += p;
  }
}

// dZ0 = ( LNbwd_noaffine(dxh) + dXs ) * (1 - X^2)     (Modules.py:519-521 backward, :270 tanh'); 16 lanes per row
__global__ __launch_bounds__(256) void enc128_lnhat_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dxh, const float* __restrict__ dXs,
                                                               float* __restrict__ dZ0, const int32_t* __restrict__ count) {
  const int T = count[0];
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c8 = (threadIdx.x & 15) * 8;
  if (t >= T) return;
  const V8 x = ld8(X + t * kD + c8);
  float rs;
  const V8 xh = ln_row8(x, 1.f, &rs);
  const V8 d = ld8(dxh + t * kD + c8);
  const f2 s2 = (d.a + d.b) + (d.c + d.d);
  const float a = group_sum16_dpp(s2.x + s2.y) * (1.f / kD);
  const float b = group_sum16_dpp(dot8(d, xh)) * (1.f / kD);
  const V8 s = ld8(dXs + t * kD + c8);
  const f2 aa = {a, a}, bb = {b, b}, rr = {rs, rs}, one = {1.f, 1.f};
  V8 o;
  o.a = (rr * (d.a - aa - xh.a * bb) + s.a) * (one - x.a * x.a);
  o.b = (rr * (d.b - aa - xh.b * bb) + s.b) * (one - x.b * x.b);
  o.c = (rr * (d.c - aa - xh.c * bb) + s.c) * (one - x.c * x.c);
  o.d = (rr * (d.d - aa - xh.d * bb) + s.d) * (one - x.d * x.d);
  st8(dZ0 + t * kD + c8, o);
}

int ml_of(int L) { return L <= 2 ? 2 : (L <= 6 ? L : 8); }
int cu_count() { return device_cu_count(); }

struct WsView { float* fold; float* bvec; u32x4* frag; float* wslab; float* red; float* colpart; float* dc; };
WsView ws_view(float* ws) {
  WsView v;
  v.fold = ws;
  v.bvec = v.fold + (size_t)MATCHA_N_HEAD * 2 * kD * kD;
  v.frag = reinterpret_cast<u32x4*>(v.bvec + 10 * kD);                   // (16-byte aligned: every size above is a multiple of 4 floats)
  v.wslab = reinterpret_cast<float*>(v.frag + (size_t)MATCHA_N_HEAD * kFragHead);
  v.red = v.wslab + (size_t)MATCHA_N_HEAD * kMaxChunks * kSlab;
  v.colpart = v.red + (size_t)MATCHA_N_HEAD * kSlab;
  v.dc = v.colpart + (size_t)kColBlocks * kD;
  return v;
}

}  // namespace

bool enc128_shape(int d) { return d == kD; }
size_t enc128_ws_floats() {
  return (size_t)MATCHA_N_HEAD * 2 * kD * kD + 10 * kD + (size_t)MATCHA_N_HEAD * kFragHead * 4 + (size_t)MATCHA_N_HEAD * kMaxChunks * kSlab +
         (size_t)MATCHA_N_HEAD * kSlab + (size_t)kColBlocks * kD + kD;
}
size_t enc128_rec_floats(const Ragged& rg) { return (size_t)rg.nhalves * MATCHA_N_HEAD * kRec; }

// lwB / lwM: the merged matrices model.hip built for this step (merged_weights); Y: dropout(fc1(attention)) . non_pad, [Tn, 128]
int launch_enc128_fwd(const matcha_tensors& p, const float* lwB, const float* lwM, const float* X, const Ragged& rg, int64_t B, int L, float* Y, float* rec,
                      float* ws, const int32_t* tok_slot, const uint64_t* seed, float p_drop, hipStream_t st) {
  const WsView v = ws_view(ws);
  {
    PrepArgs a;
    a.lwB = lwB; a.lwM = lwM; a.gq = p.ln_q_g; a.bq = p.ln_q_b; a.gk = p.ln_k_g; a.gv = p.ln_v_g; a.bv = p.ln_v_b; a.fc1_b = p.fc1_b;
    a.fold = v.fold; a.frag = v.frag; a.bvec = v.bvec;
    hipLaunchKernelGGL(enc128_prep_kernel, dim3(128 + 9), dim3(256), 0, st, a);
    MATCHA_CHECK_LAUNCH("enc128_prep_kernel");
  }
  FwdArgs g;
  g.X = X; g.count = rg.count; g.half_meta = rg.half_meta; g.tok_pos = rg.tok_pos; g.tok_slot = tok_slot; g.L = L; g.nhalves = rg.nhalves;
  // one workgroup per CU, each with an equal share of the half tiles (the count is on the device): one round of workgroups, and the eight
  // weight reloads (196 KB each) are amortised over the whole share
  int nwg = cu_count();
  if (nwg > rg.nhalves) nwg = rg.nhalves > 0 ? rg.nhalves : 1;
  g.nwg = nwg;
  g.frag = v.frag; g.bvec = v.bvec; g.Y = Y; g.rec = rec; g.seed = seed; g.p_drop = p_drop;
  auto launch = [&](auto kfn) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kfn, dim3(nwg > 0 ? nwg : 1), dim3(512), kFwdLdsBytes, st, g);
  };
  // algorithmic flops: the reference's formulation -- 8 heads x 4 GEMMs of 2 d^2 per token forward (this kernel executes half of them)
  ProfScope ps(MATCHA_PROF_FUSED_FWD, (double)(B * L + 1) * MATCHA_N_HEAD * 4.0 * 2.0 * kD * kD, st);
  switch (ml_of(L)) {
    case 2: launch(enc128_fwd_kernel<2>); break;
    case 3: launch(enc128_fwd_kernel<3>); break;
    case 4: launch(enc128_fwd_kernel<4>); break;
    case 5: launch(enc128_fwd_kernel<5>); break;
    case 6: launch(enc128_fwd_kernel<6>); break;
    default: launch(enc128_fwd_kernel<8>); break;
  }
  MATCHA_CHECK_LAUNCH("enc128_fwd_kernel");
  return MATCHA_OK;
}

// dDyn = dL/d(fc1 output before the bias); dxh [Tn, 128] scratch (zeroed here); lwdB / lwdM receive dB_all / dM_all for merged_chain;
// the LayerNorm affine and fc1 bias gradients are ACCUMULATED into grads; dZ0 = gradient at the next_w pre-activation
int launch_enc128_bwd(const matcha_tensors& p, const float* lwB, const float* lwM, const float* X, const float* dDyn, const float* dXs, const Ragged& rg,
                      int64_t B, int L, float* dxh, const float* rec, float* ws, float* lwdB, float* lwdM, matcha_tensors& grads, float* dZ0,
                      hipStream_t st) {
  const WsView v = ws_view(ws);
  const int64_t tcap = B * L + 1;
  MATCHA_TRY(zero_async(dxh, (size_t)tcap * kD * sizeof(float), st));
  hipLaunchKernelGGL(enc128_colsum_kernel, dim3(kColBlocks), dim3(256), 0, st, dDyn, rg.count, v.colpart);
  MATCHA_CHECK_LAUNCH("enc128_colsum_kernel");
  int nchunks = kMaxChunks;
  if (nchunks > rg.nhalves) nchunks = rg.nhalves > 0 ? rg.nhalves : 1;
  {
    BwdArgs g;
    g.X = X; g.dDyn = dDyn; g.count = rg.count; g.half_meta = rg.half_meta; g.tok_pos = rg.tok_pos; g.L = L; g.nhalves = rg.nhalves; g.nchunks = nchunks;
    g.fold = v.fold; g.dxh = dxh; g.wslab = v.wslab; g.rec = rec;
    auto launch = [&](auto kfn) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kfn, dim3(MATCHA_N_HEAD * nchunks), dim3(512), kBwdLdsBytes, st, g);
    };
    ProfScope ps(MATCHA_PROF_FUSED_BWD, (double)tcap * MATCHA_N_HEAD * 8.0 * 2.0 * kD * kD, st);
    switch (ml_of(L)) {
      case 2: launch(enc128_bwd_kernel<2>); break;
      case 3: launch(enc128_bwd_kernel<3>); break;
      case 4: launch(enc128_bwd_kernel<4>); break;
      case 5: launch(enc128_bwd_kernel<5>); break;
      case 6: launch(enc128_bwd_kernel<6>); break;
      default: launch(enc128_bwd_kernel<8>); break;
    }
    MATCHA_CHECK_LAUNCH("enc128_bwd_kernel");
  }
  {
    ReduceArgs a;
    a.wslab = v.wslab; a.nchunks = nchunks; a.red = v.red; a.colpart = v.colpart; a.dc = v.dc;
    hipLaunchKernelGGL(enc128_reduce_kernel, dim3((unsigned)cdiv(kSlab, 256), MATCHA_N_HEAD + 1), dim3(256), 0, st, a);
    MATCHA_CHECK_LAUNCH("enc128_reduce_kernel");
  }
  {
    UnfoldArgs a;
    a.red = v.red; a.dc = v.dc; a.lwB = lwB; a.lwM = lwM; a.gq = p.ln_q_g; a.bq = p.ln_q_b; a.gk = p.ln_k_g; a.gv = p.ln_v_g; a.bv = p.ln_v_b;
    a.lwdB = lwdB; a.lwdM = lwdM; a.dgq = grads.ln_q_g; a.dbq = grads.ln_q_b; a.dgk = grads.ln_k_g; a.dgv = grads.ln_v_g; a.dbv = grads.ln_v_b;
    a.dfc1_b = grads.fc1_b; a.dxh = dxh; a.count = rg.count;
    hipLaunchKernelGGL(enc128_unfold_kernel, dim3(kD), dim3(256), 0, st, a);
    MATCHA_CHECK_LAUNCH("enc128_unfold_kernel");
  }
  hipLaunchKernelGGL(enc128_lnhat_bwd_kernel, dim3((unsigned)cdiv(tcap, 16)), dim3(256), 0, st, X, dxh, dXs, dZ0, rg.count);
  MATCHA_CHECK_LAUNCH("enc128_lnhat_bwd_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
