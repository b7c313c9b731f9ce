// Wide-tile GEMMs for the layer-by-layer path at embed_dim >= 128 (BASELINE configs[3], [4]): the Q/K/V and fc1
// projections (NT: Y = X.W^T, Modules.py:527-529, :572), their input gradients (NN: dX = dY.W) and every weight gradient
// (TN: dW = dY^T.X).
//
// Round 5: fp32-ACCURATE products on the bf16 matrix pipe.  The f32 MFMA (v_mfma_f32_32x32x2_f32) runs at the f32 vector rate and on the
// vector ALUs; here every operand fragment -- eight consecutive contraction indices of one row, read from the f32 staging tile -- is split
// in registers into three bf16 planes (v = h + m + l to 2^-27, round to nearest) and a 32 x 32 x 16 block is the six plane products above
// 2^-26 (Al Bh + Ah Bl + Am Bm + Am Bh + Ah Bm + Ah Bh, one v_mfma_f32_32x32x16_bf16 each, f32 accumulate): 6 x 32 cycles on the matrix pipe
// instead of 8 x 64 on the vector ALUs; the result differs from an f32 fma chain by ~1e-7 relative, that chain's own rounding level
// (tests/test_cpu_bf16x3.py; fused_fwd32.hip has the longer note).  Staging, tiling, pipeline and epilogues are unchanged:
//
//   workgroup = 256 threads = 4 waves in a 2 x 2 grid; output tile 128 x 128; each wave owns 64 x 64 = FOUR 32x32 accumulators,
//   so one 16-byte LDS read per operand row feeds twice the MFMAs of the 32 x 64 wave tile of gemm_lds.hip (4 reads per 16
//   MFMAs instead of 3 per 8) and the address arithmetic is amortised over twice the work -- on this part every VALU instruction
//   costs its issue slots next to the f32 MFMAs (tools/ubench/mfma_valu.hip);
//   contraction in 32-deep chunks, double-buffered in LDS (2 x 36 KB -> two workgroups per CU), global -> register loads of chunk
//   s + 1 in flight during the 64 MFMAs per wave of chunk s, one barrier per chunk;
//   XCD-aware tile order: the workgroups that share an activation row tile (NT/NN) or a token partition (TN) are consecutive
//   ON ONE XCD (blockIdx round-robins over the 8 XCDs), so the shared operand is fetched into that XCD's L2 once instead of
//   eight times.
// TN stages [32 tokens][128 columns] slabs of both operands with fully coalesced 512-byte row reads (gemm_tn_kernel reads 4-byte
// elements straight from global memory: 4 x the requests) and keeps the fixed-order slab reduction of gemm_f32.hip.
//
// Eligibility (launch_*_wide return false otherwise and the caller falls back to the 64-wide kernels): N % 128 == 0,
// K % 32 == 0, 16-byte aligned operands, no row maps / grouped mode (the adj front end keeps gemm_lds.hip).
#include "kernels.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kWM = 128, kWN = 128, kWK = 32;
constexpr int kLdK = kWK + 4;          // k-major tiles: [row][k], 36 floats per row (conflict-free 16-byte reads, gemm_lds.hip)
constexpr int kLdN = kWN + 4;          // n-major tiles: [k][n], 132 floats per row

// eight f32 values (one lane's contraction slots of a 32x32x16 fragment) as three bf16 planes
struct Frag3 { u32x4 h, m, l; };
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  const f2 v = {a, b};
  const bf16x2 hb = __builtin_convertvector(v, bf16x2);                 // v_cvt_pk_bf16_f32: round to nearest even
  const f2 r1 = v - __builtin_convertvector(hb, f2);
  const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
  const f2 r2 = r1 - __builtin_convertvector(mb, f2);
  const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
  h = __builtin_bit_cast(uint32_t, hb); m = __builtin_bit_cast(uint32_t, mb); l = __builtin_bit_cast(uint32_t, lb);
}
__device__ __forceinline__ Frag3 split8(const float* v) {
  Frag3 f;
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], m[q], l[q]);
  f.h = (u32x4){h[0], h[1], h[2], h[3]}; f.m = (u32x4){m[0], m[1], m[2], m[3]}; f.l = (u32x4){l[0], l[1], l[2], l[3]};
  return f;
}
#define WIDE_BF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)), __builtin_bit_cast(bf16x8, (B)), (C), 0, 0, 0)
// acc += A . B over 16 contraction indices: the six plane products, smallest first
__device__ __forceinline__ void mma6(f32x16& acc, const Frag3& a, const Frag3& b) {
  acc = WIDE_BF(a.l, b.h, acc); acc = WIDE_BF(a.h, b.l, acc); acc = WIDE_BF(a.m, b.m, acc);
  acc = WIDE_BF(a.m, b.h, acc); acc = WIDE_BF(a.h, b.m, acc); acc = WIDE_BF(a.h, b.h, acc);
}

template <int FLAGS>
struct Epi {
  __device__ __forceinline__ static bool has(const GemmArgs& g, int f) { return FLAGS >= 0 ? (FLAGS & f) != 0 : (g.flags & f) != 0; }
};

// One workgroup walks `ntpb` consecutive tiles of the row-major tile list (the column tiles of a row tile first) as ONE software
// pipeline over (tile, k chunk) steps: the operands of step s + 1 are on their way from global memory during the 64 MFMAs per
// wave of step s -- also across a tile boundary, so a tile's prologue latency and its epilogue are paid once per workgroup,
// not once per 256 MFMAs (at K = 128 a tile is only four chunks: the one-tile-per-workgroup version sat at 75 TFLOP/s).
// XCD-aware order: workgroup ids b, b + 8, ... run on one XCD; each XCD gets a contiguous range of the tile list, so the
// workgroups that share an activation row tile hit the same L2.
// in-situ ablations (wrong results on purpose; bit 0 no epilogue, bit 1 no MFMAs, bit 2 no prefetch): a COMPILE-time switch (-DWIDE_ABL=n, like
// ENC_ABL / F32_ABL) -- it used to be the run-time option fused_dbg, whose bit 0 has a documented meaning elsewhere
#ifndef WIDE_ABL
#define WIDE_ABL 0
#endif
constexpr int kWideAbl = WIDE_ABL;
template <bool B_KN, int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(GemmArgs g, int64_t nx, int64_t ny, int ntpb, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int A_SZ = kWM * kLdK;
  constexpr int B_SZ = B_KN ? kWK * kLdN : kWN * kLdK;
  constexpr int BUF = A_SZ + B_SZ;                        // one pipeline stage: [A | B], contiguous (the epilogue borrows the idle stage)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;
  const int z = blockIdx.y;
  const float* __restrict__ A = g.A[z];
  const float* __restrict__ B = g.B[z];
  float* __restrict__ C = g.C[z];
  // the row count may live on the device (ragged layout: the launch is sized for B*L + 1 rows, ~30 % more than the real tokens):
  // the tile list -- and with it the share of every XCD -- is built from the REAL rows, or two of the eight XCDs would idle
  const int64_t M = g.m_dev ? (int64_t)(*g.m_dev) : g.M;
  const int64_t K = g.K;
  ny = (M + kWM - 1) / kWM;
  const int64_t total = nx * ny;
  const int64_t nwg = (total + ntpb - 1) / ntpb;
  const int64_t per = (nwg + 7) / 8;
  const int64_t wg = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int64_t)(blockIdx.x >> 3) >= per || wg >= nwg) return;
  const int64_t t_first = wg * ntpb;
  int n_my = ntpb;
  if (t_first + n_my > total) n_my = (int)(total - t_first);
  if (n_my <= 0) return;
  // staging maps: k-major tiles: thread -> (row = tid / 8 + 32 i, float4 column tid % 8); n-major B: (k row = tid / 32 + 8 i, float4 column tid % 32)
  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  const int nrow = tid >> 5, nc4 = (tid & 31) * 4;
  const float *ap0, *ap1, *ap2, *ap3, *bp0, *bp1, *bp2, *bp3;
  const int64_t bstep = B_KN ? g.ldb : 1;                 // B advances by kc rows (n-major tile) or kc columns (k-major tile)
#define WIDE_PTRS(T)                                                                          \
  do {                                                                                        \
    const int64_t ty__ = (T) / nx, tx__ = (T) - ty__ * nx;                                    \
    const int64_t m0__ = ty__ * kWM, n0__ = tx__ * kWN;                                       \
    int64_t gm__;                                                                             \
    gm__ = m0__ + srow;      gm__ = gm__ < M ? gm__ : M - 1; ap0 = A + gm__ * g.lda + sc4;    /* clamped: rows past M are computed, never stored */ \
    gm__ = m0__ + srow + 32; gm__ = gm__ < M ? gm__ : M - 1; ap1 = A + gm__ * g.lda + sc4;    \
    gm__ = m0__ + srow + 64; gm__ = gm__ < M ? gm__ : M - 1; ap2 = A + gm__ * g.lda + sc4;    \
    gm__ = m0__ + srow + 96; gm__ = gm__ < M ? gm__ : M - 1; ap3 = A + gm__ * g.lda + sc4;    \
    if (B_KN) {                                                                               \
      bp0 = B + (int64_t)nrow * g.ldb + n0__ + nc4; bp1 = bp0 + 8 * g.ldb; bp2 = bp0 + 16 * g.ldb; bp3 = bp0 + 24 * g.ldb; \
    } else {                                                                                  \
      bp0 = B + (n0__ + srow) * g.ldb + sc4; bp1 = bp0 + 32 * g.ldb; bp2 = bp0 + 64 * g.ldb; bp3 = bp0 + 96 * g.ldb;       \
    }                                                                                         \
  } while (0)
  // (macros, not lambdas: register arrays captured by reference end up in scratch memory -- fused_fwd.hip has the same note)
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define WIDE_GLOAD(KC)                                                                        \
  do {                                                                                        \
    const int64_t kc__ = (KC);                                                                \
    ra0 = *reinterpret_cast<const float4*>(ap0 + kc__); ra1 = *reinterpret_cast<const float4*>(ap1 + kc__);   \
    ra2 = *reinterpret_cast<const float4*>(ap2 + kc__); ra3 = *reinterpret_cast<const float4*>(ap3 + kc__);   \
    rb0 = *reinterpret_cast<const float4*>(bp0 + kc__ * bstep); rb1 = *reinterpret_cast<const float4*>(bp1 + kc__ * bstep); \
    rb2 = *reinterpret_cast<const float4*>(bp2 + kc__ * bstep); rb3 = *reinterpret_cast<const float4*>(bp3 + kc__ * bstep); \
  } while (0)
#define WIDE_LSTORE(STAGE)                                                                    \
  do {                                                                                        \
    float* as__ = (STAGE) + srow * kLdK + sc4;                                                \
    *reinterpret_cast<float4*>(as__) = ra0; *reinterpret_cast<float4*>(as__ + 32 * kLdK) = ra1;               \
    *reinterpret_cast<float4*>(as__ + 64 * kLdK) = ra2; *reinterpret_cast<float4*>(as__ + 96 * kLdK) = ra3;   \
    if (!B_KN) {                                                                              \
      float* bs__ = (STAGE) + A_SZ + srow * kLdK + sc4;                                       \
      *reinterpret_cast<float4*>(bs__) = rb0; *reinterpret_cast<float4*>(bs__ + 32 * kLdK) = rb1;             \
      *reinterpret_cast<float4*>(bs__ + 64 * kLdK) = rb2; *reinterpret_cast<float4*>(bs__ + 96 * kLdK) = rb3; \
    } else {                                                                                  \
      float* bs__ = (STAGE) + A_SZ + nrow * kLdN + nc4;                                       \
      *reinterpret_cast<float4*>(bs__) = rb0; *reinterpret_cast<float4*>(bs__ + 8 * kLdN) = rb1;              \
      *reinterpret_cast<float4*>(bs__ + 16 * kLdN) = rb2; *reinterpret_cast<float4*>(bs__ + 24 * kLdN) = rb3; \
    }                                                                                         \
  } while (0)
  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (Epi<FLAGS>::has(g, MATCHA_EPI_DROPOUT)) {
    key = rng_key(*g.seed, g.stream_id);
    thr = dropout_threshold(g.p_drop);
    keep_scale = 1.f / (1.f - g.p_drop);
  }
  const int ec4 = (lane & 15) * 4, er = lane >> 4;        // epilogue read-back: 16 lanes x 16 bytes = one 256-byte row segment
  f32x16 acc[2][2];
  const int nkc = (int)(K / kWK);
  const int nsteps = n_my * nkc;
  WIDE_PTRS(t_first);
  WIDE_GLOAD(0);
  WIDE_LSTORE(lds);
  __syncthreads();
  int kci = 0;
  int64_t tile = t_first;
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more && !(dbg & 4)) {                              // operands of the next step (possibly the next tile's first chunk): in flight during this step's MFMAs
      if (kci + 1 == nkc) { WIDE_PTRS(tile + 1); WIDE_GLOAD(0); }
      else WIDE_GLOAD((int64_t)(kci + 1) * kWK);
    }
    if (kci == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    }
    const float* As = lds + (s & 1) * BUF;
    const float* Bs = As + A_SZ;
    // lane (r, h) supplies contraction indices 16 s2 + 8 h + {0..7} of its row to both operands of a 32x32x16 step
    const float* a0p = &As[(64 * wr + r) * kLdK + 8 * h];
    const float* a1p = a0p + 32 * kLdK;
    if (!(dbg & 2))
#pragma unroll
    for (int s2 = 0; s2 < kWK / 16; ++s2) {
      float va0[8], va1[8], vb0[8], vb1[8];
      {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(a0p + 16 * s2), x1 = *reinterpret_cast<const f32x4*>(a0p + 16 * s2 + 4);
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(a1p + 16 * s2), y1 = *reinterpret_cast<const f32x4*>(a1p + 16 * s2 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { va0[j] = x0[j]; va0[4 + j] = x1[j]; va1[j] = y0[j]; va1[4 + j] = y1[j]; }
      }
      if (!B_KN) {
        const float* b0p = &Bs[(64 * wc + r) * kLdK + 16 * s2 + 8 * h];
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(b0p), x1 = *reinterpret_cast<const f32x4*>(b0p + 4);
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(b0p + 32 * kLdK), y1 = *reinterpret_cast<const f32x4*>(b0p + 32 * kLdK + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { vb0[j] = x0[j]; vb0[4 + j] = x1[j]; vb1[j] = y0[j]; vb1[4 + j] = y1[j]; }
      } else {
        const float* q = &Bs[(16 * s2 + 8 * h) * kLdN + 64 * wc + r];
#pragma unroll
        for (int j = 0; j < 8; ++j) { vb0[j] = q[j * kLdN]; vb1[j] = q[j * kLdN + 32]; }
      }
      const Frag3 fa0 = split8(va0), fa1 = split8(va1), fb0 = split8(vb0), fb1 = split8(vb1);
      mma6(acc[0][0], fa0, fb0); mma6(acc[0][1], fa0, fb1); mma6(acc[1][0], fa1, fb0); mma6(acc[1][1], fa1, fb1);
    }
    float* const idle = lds + ((s + 1) & 1) * BUF;        // the stage the next step's operands will be written to
    if (kci + 1 == nkc && (dbg & 1)) { kci = 0; ++tile; }
    else if (kci + 1 == nkc) {
      // ---- epilogue of this tile: bias -> tanh -> +residual -> dropout -> row mask -> *(1 - (aux*aux_scale)^2) -> (+=) ----
      // The MFMA accumulator layout gives a lane ONE column and 16 rows: written as it stands that is 64 four-byte stores per
      // lane, and store ISSUE (not bandwidth) is what they cost.  Each wave parks half of its 64 x 64 tile (32 rows) in its
      // quarter of the IDLE pipeline stage (every wave passed the barrier behind the step that last read it) and reads it back
      // row-major: 16 lanes x 16 bytes = a 256-byte row segment per quarter wave; residual / aux / accumulate operands are
      // read the same way.  A wave reads back only what it wrote itself (a wave's LDS operations execute in order).
      constexpr int kLdE = 68;
      float* Es = idle + wave * (32 * kLdE);               // 4 x 8.7 KB <= one stage (36.0 / 34.5 KB)
      const int64_t ty = tile / nx, tx = tile - ty * nx;
      const int64_t m0 = ty * kWM, col = tx * kWN + 64 * wc + ec4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (Epi<FLAGS>::has(g, MATCHA_EPI_BIAS)) bv = *reinterpret_cast<const float4*>(g.bias[z] + col);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) Es[((reg & 3) + 8 * (reg >> 2) + 4 * h) * kLdE + 32 * j + r] = acc[i][j][reg];
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
          const int lrow = 4 * it + er;
          const int64_t row = m0 + 64 * wr + 32 * i + lrow;
          if (row >= M) continue;
          float4 v = *reinterpret_cast<const float4*>(&Es[lrow * kLdE + ec4]);
          const int64_t off = row * g.ldc + col;
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          if (Epi<FLAGS>::has(g, MATCHA_EPI_TANH)) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
          if (Epi<FLAGS>::has(g, MATCHA_EPI_RESIDUAL)) {
            const float4 q = *reinterpret_cast<const float4*>(g.residual + off);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
          }
          if (Epi<FLAGS>::has(g, MATCHA_EPI_DROPOUT)) {
            const uint32_t crow = g.rng_row_map ? (uint32_t)g.rng_row_map[row] : (uint32_t)row;
            v.x = (rng_u32(key, crow, (uint32_t)col) >= thr) ? v.x * keep_scale : 0.f;
            v.y = (rng_u32(key, crow, (uint32_t)col + 1u) >= thr) ? v.y * keep_scale : 0.f;
            v.z = (rng_u32(key, crow, (uint32_t)col + 2u) >= thr) ? v.z * keep_scale : 0.f;
            v.w = (rng_u32(key, crow, (uint32_t)col + 3u) >= thr) ? v.w * keep_scale : 0.f;
          }
          if (Epi<FLAGS>::has(g, MATCHA_EPI_ROWMASK)) {
            if (g.row_ids[row] == 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          if (Epi<FLAGS>::has(g, MATCHA_EPI_DTANH)) {
            const float4 q = *reinterpret_cast<const float4*>(g.aux + off);
            const float a0 = q.x * g.aux_scale, a1 = q.y * g.aux_scale, a2 = q.z * g.aux_scale, a3 = q.w * g.aux_scale;
            v.x *= (1.f - a0 * a0); v.y *= (1.f - a1 * a1); v.z *= (1.f - a2 * a2); v.w *= (1.f - a3 * a3);
          }
          if (Epi<FLAGS>::has(g, MATCHA_EPI_ACCUM)) {
            const float4 q = *reinterpret_cast<const float4*>(C + off);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
          }
          *reinterpret_cast<float4*>(C + off) = v;
        }
      }
      if (more) __syncthreads();                           // every wave is done with its quarter before the stage is refilled
      kci = 0;
      ++tile;
    } else {
      ++kci;
    }
    if (more) WIDE_LSTORE(idle);
    __syncthreads();
  }
#undef WIDE_PTRS
#undef WIDE_GLOAD
#undef WIDE_LSTORE
}

template <bool B_KN, int FLAGS>
void launch_wide_one(const GemmArgs& g, hipStream_t st) {
  auto kfn = gemm_wide_kernel<B_KN, FLAGS>;
  const size_t lds = (size_t)(2 * kWM * kLdK + 2 * (B_KN ? kWK * kLdN : kWN * kLdK)) * sizeof(float);
  static bool configured = false;       // per instantiation
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  const int64_t nx = g.N / kWN, ny = cdiv(g.M, kWM);
  // tiles per workgroup: long pipelines, but at least ~4 workgroups per CU slot (512 slots) so that the tail stays short
  int64_t ntpb = (nx * ny * g.batch) / 2048;
  if (ntpb > 8) ntpb = 8;
  if (ntpb < 1) ntpb = 1;
  const int64_t nwg = cdiv(nx * ny, ntpb);
  hipLaunchKernelGGL(kfn, dim3((unsigned)(cdiv(nwg, 8) * 8), (unsigned)g.batch), dim3(256), lds, st, g, nx, ny, (int)ntpb, kWideAbl);
}

// ---- TN: C[M,N] partial of one token partition = A[rows, M]^T . B[rows, N] -----------------------------------------------------
struct WideTnArgs {
  const float* A; const float* B;
  float* slab; float* colslab;
  int64_t M, N, R, lda, ldb, slab_stride, rows_per_block;
  const int32_t* r_dev;
  int tiles_m, tiles_n, P;
};

__global__ __launch_bounds__(256, 2) void gemm_tn_wide_kernel(WideTnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int T_SZ = kWK * kLdN;                 // [32 tokens][132]
  float* const As0 = lds;
  float* const As1 = lds + T_SZ;
  float* const Bs0 = lds + 2 * T_SZ;
  float* const Bs1 = lds + 3 * T_SZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;
  // XCD-aware order: the tiles of one token partition are consecutive on one XCD (they all read the partition's rows)
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
  const int64_t total = tiles * g.P;
  const int64_t per = (total + 7) / 8;
  const int64_t t = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int64_t)(blockIdx.x >> 3) >= per || t >= total) return;
  const int64_t p = t / tiles, tile = t - p * tiles;
  const int64_t mo0 = (tile / g.tiles_n) * kWM, no0 = (tile % g.tiles_n) * kWN;
  const int64_t R = g.r_dev ? (int64_t)(*g.r_dev) : g.R;
  int64_t rpb = g.rows_per_block;
  if (g.r_dev) {                                   // split the ACTUAL rows evenly over the launched partitions
    rpb = (R + g.P - 1) / g.P;
    rpb = (rpb + kWK - 1) / kWK * kWK;
    if (rpb < kWK) rpb = kWK;
  }
  const int64_t rbeg = p * rpb;
  int64_t rend = rbeg + rpb; if (rend > R) rend = R;
  if (rend < rbeg) rend = rbeg;                    // empty partition: writes a zero slab
  const int nrow = tid >> 5, nc4 = (tid & 31) * 4;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const float* const Ab = g.A + mo0 + nc4;
  const float* const Bb = g.B + no0 + nc4;
  const int64_t rlast = rend > 0 ? rend - 1 : 0;
#define TNW_ROW(RA, RB, I, R0)                                                                \
  do {                                                                                        \
    const int64_t rr__ = (R0) + nrow + 8 * (I);                                               \
    const int64_t rc__ = rr__ < rend ? rr__ : rlast;      /* clamped load, zeroed by the mask: no predicated loads in hot loops */ \
    const float m__ = rr__ < rend ? 1.f : 0.f;                                                \
    const float4 a__ = *reinterpret_cast<const float4*>(Ab + rc__ * g.lda);                   \
    RB = *reinterpret_cast<const float4*>(Bb + rc__ * g.ldb);                                 \
    RA = make_float4(a__.x * m__, a__.y * m__, a__.z * m__, a__.w * m__);                     \
  } while (0)
#define TNW_GLOAD(R0) do { TNW_ROW(ra0, rb0, 0, R0); TNW_ROW(ra1, rb1, 1, R0); TNW_ROW(ra2, rb2, 2, R0); TNW_ROW(ra3, rb3, 3, R0); } while (0)
#define TNW_LSTORE(AS, BS)                                                                    \
  do {                                                                                        \
    float* as__ = (AS) + nrow * kLdN + nc4;                                                   \
    float* bs__ = (BS) + nrow * kLdN + nc4;                                                   \
    *reinterpret_cast<float4*>(as__) = ra0; *reinterpret_cast<float4*>(as__ + 8 * kLdN) = ra1;                \
    *reinterpret_cast<float4*>(as__ + 16 * kLdN) = ra2; *reinterpret_cast<float4*>(as__ + 24 * kLdN) = ra3;   \
    *reinterpret_cast<float4*>(bs__) = rb0; *reinterpret_cast<float4*>(bs__ + 8 * kLdN) = rb1;                \
    *reinterpret_cast<float4*>(bs__ + 16 * kLdN) = rb2; *reinterpret_cast<float4*>(bs__ + 24 * kLdN) = rb3;   \
  } while (0)
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
  float cs0 = 0.f, cs1 = 0.f;
  const int64_t nchunks = (rend - rbeg + kWK - 1) / kWK;
  if (nchunks > 0) {
    TNW_GLOAD(rbeg);
    TNW_LSTORE(As0, Bs0);
  }
  __syncthreads();
  for (int64_t s = 0; s < nchunks; ++s) {
    const bool more = s + 1 < nchunks;
    if (more) TNW_GLOAD(rbeg + (s + 1) * kWK);
    const float* As = (s & 1) ? As1 : As0;
    const float* Bs = (s & 1) ? Bs1 : Bs0;
    // lane (r, h) supplies tokens 16 s2 + 8 h + {0..7} of its column to both operands of a 32x32x16 step
    const float* apx = &As[(8 * h) * kLdN + 64 * wr + r];
    const float* bpx = &Bs[(8 * h) * kLdN + 64 * wc + r];
#pragma unroll
    for (int s2 = 0; s2 < kWK / 16; ++s2) {
      float va0[8], va1[8], vb0[8], vb1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        va0[j] = apx[(16 * s2 + j) * kLdN]; va1[j] = apx[(16 * s2 + j) * kLdN + 32];
        vb0[j] = bpx[(16 * s2 + j) * kLdN]; vb1[j] = bpx[(16 * s2 + j) * kLdN + 32];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { cs0 += va0[j]; cs1 += va1[j]; }
      const Frag3 fa0 = split8(va0), fa1 = split8(va1), fb0 = split8(vb0), fb1 = split8(vb1);
      mma6(acc[0][0], fa0, fb0); mma6(acc[0][1], fa0, fb1); mma6(acc[1][0], fa1, fb0); mma6(acc[1][1], fa1, fb1);
    }
    if (more) TNW_LSTORE((s & 1) ? As0 : As1, (s & 1) ? Bs0 : Bs1);
    __syncthreads();
  }
#undef TNW_ROW
#undef TNW_GLOAD
#undef TNW_LSTORE
  float* slab = g.slab + p * g.slab_stride;
  {
    // row-major 16-byte stores through the idle staging LDS (see gemm_wide_kernel's epilogue)
    constexpr int kLdE = 68;
    float* Es = lds + wave * 64 * kLdE;            // 4 x 17 KB = the four staging tiles' 67.6 KB
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) Es[(32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h) * kLdE + 32 * j + r] = acc[i][j][reg];
    const int ec4 = (lane & 15) * 4, er = lane >> 4;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int lrow = 4 * it + er;
      *reinterpret_cast<float4*>(slab + (mo0 + 64 * wr + lrow) * g.N + no0 + 64 * wc + ec4) = *reinterpret_cast<const float4*>(&Es[lrow * kLdE + ec4]);
    }
  }
  if (g.colslab && no0 == 0 && wc == 0) {           // column sums of A (bias gradients): both token parities, one writer per column
    cs0 += __shfl_xor(cs0, 32, kWave);
    cs1 += __shfl_xor(cs1, 32, kWave);
    if (h == 0) {
      g.colslab[p * g.slab_stride + mo0 + 64 * wr + r] = cs0;
      g.colslab[p * g.slab_stride + mo0 + 64 * wr + 32 + r] = cs1;
    }
  }
}

void tn_wide_partition(int64_t M, int64_t N, int64_t R, int* tm, int* tn, int* P, int64_t* rpb) {
  *tm = (int)(M / kWM);
  *tn = (int)(N / kWN);
  const int64_t tiles = (int64_t)(*tm) * (*tn);
  int64_t want = cdiv(512, tiles);                  // two workgroups per CU
  const int64_t maxp = cdiv(R, 4 * kWK);
  int64_t p = want < maxp ? want : maxp;
  if (p < 1) p = 1;
  *rpb = cdiv(cdiv(R, p), kWK) * kWK;
  *P = (int)cdiv(R, *rpb);
  if (*P < 1) *P = 1;
}

}  // namespace

bool gemm_wide_eligible(bool b_kn, const GemmArgs& g) {
  if (options().disable_wide_gemm) return false;
  if (g.a_row_map || g.c_row_map || g.seg) return false;
  if (g.N % kWN != 0 || g.K % kWK != 0 || g.K < 2 * kWK || g.M < 1) return false;
  if (g.lda % 4 != 0 || g.ldb % 4 != 0 || g.lda < g.K) return false;
  if (!b_kn && g.ldb < g.K) return false;
  if (b_kn && g.ldb < g.N) return false;
  if (g.ldc % 4 != 0) return false;                        // the epilogue stores (and reads residual / aux / C) 16 bytes at a time
  for (int z = 0; z < g.batch; ++z)
    if (((uintptr_t)g.A[z]) % 16 != 0 || ((uintptr_t)g.B[z]) % 16 != 0 || ((uintptr_t)g.C[z]) % 16 != 0 ||
        (g.bias[z] && ((uintptr_t)g.bias[z]) % 16 != 0)) return false;
  if ((g.residual && ((uintptr_t)g.residual) % 16 != 0) || (g.aux && ((uintptr_t)g.aux) % 16 != 0)) return false;
  return true;
}

int launch_gemm_wide(bool b_kn, const GemmArgs& g, hipStream_t st) {
  ProfScope ps(b_kn ? MATCHA_PROF_GEMM_NN : MATCHA_PROF_GEMM_NT, 2.0 * (double)g.M * (double)g.N * (double)g.K * g.batch, st);
  const int F = g.flags;
  constexpr int B_ = MATCHA_EPI_BIAS, T_ = MATCHA_EPI_TANH, R_ = MATCHA_EPI_RESIDUAL, D_ = MATCHA_EPI_DROPOUT, M_ = MATCHA_EPI_ROWMASK,
                G_ = MATCHA_EPI_DTANH;
  if (!b_kn) {
    switch (F) {
      case 0: launch_wide_one<false, 0>(g, st); break;
      case B_: launch_wide_one<false, B_>(g, st); break;
      case B_ | T_: launch_wide_one<false, B_ | T_>(g, st); break;
      case B_ | T_ | D_: launch_wide_one<false, B_ | T_ | D_>(g, st); break;
      case B_ | M_: launch_wide_one<false, B_ | M_>(g, st); break;
      case B_ | M_ | D_: launch_wide_one<false, B_ | M_ | D_>(g, st); break;
      case B_ | R_: launch_wide_one<false, B_ | R_>(g, st); break;
      default: launch_wide_one<false, -1>(g, st); break;
    }
  } else {
    switch (F) {
      case 0: launch_wide_one<true, 0>(g, st); break;
      case G_: launch_wide_one<true, G_>(g, st); break;
      case G_ | D_: launch_wide_one<true, G_ | D_>(g, st); break;
      case R_ | M_: launch_wide_one<true, R_ | M_>(g, st); break;
      case R_ | M_ | D_: launch_wide_one<true, R_ | M_ | D_>(g, st); break;
      default: launch_wide_one<true, -1>(g, st); break;
    }
  }
  MATCHA_CHECK_LAUNCH("gemm_wide_kernel");
  return MATCHA_OK;
}

bool gemm_tn_wide_eligible(int64_t M, int64_t N, int64_t R, int64_t lda, int64_t ldb, const float* A, const float* B, const int64_t* b_gather) {
  if (options().disable_wide_gemm || b_gather) return false;
  if (M % kWM != 0 || N % kWN != 0 || R < 1) return false;
  if (lda % 4 != 0 || ldb % 4 != 0 || lda < M || ldb < N) return false;
  if (((uintptr_t)A) % 16 != 0 || ((uintptr_t)B) % 16 != 0) return false;
  return true;
}

size_t gemm_tn_wide_ws_bytes(int64_t M, int64_t N, int64_t R) {
  if (M % kWM != 0 || N % kWN != 0) return 0;
  int tm, tn, P; int64_t rpb;
  tn_wide_partition(M, N, R > 0 ? R : 1, &tm, &tn, &P, &rpb);
  return align_up((size_t)P * (size_t)(M * N + M) * sizeof(float), 256);
}

// the slab part of launch_gemm_tn for eligible shapes; the caller reduces the P slabs (launch_slab_reduce)
int launch_gemm_tn_wide(const float* A, const float* B, int64_t M, int64_t N, int64_t R, int64_t lda, int64_t ldb, bool colsum, void* ws,
                        size_t ws_bytes, const int32_t* r_dev, int* P_out, int64_t* slab_stride_out, hipStream_t st) {
  int tm, tn, P; int64_t rpb;
  tn_wide_partition(M, N, R > 0 ? R : 1, &tm, &tn, &P, &rpb);
  const size_t need = (size_t)P * (size_t)(M * N + M) * sizeof(float);
  if (ws_bytes < need) { set_error("gemm TN (wide) workspace too small: %zu < %zu", ws_bytes, need); return MATCHA_ENOMEM; }
  WideTnArgs g;
  g.A = A; g.B = B; g.slab = (float*)ws;
  g.slab_stride = M * N + (colsum ? M : 0);
  g.colslab = colsum ? (float*)ws + M * N : nullptr;
  g.M = M; g.N = N; g.R = R; g.lda = lda; g.ldb = ldb; g.rows_per_block = rpb; g.r_dev = r_dev;
  g.tiles_m = tm; g.tiles_n = tn; g.P = P;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  const size_t lds = (size_t)4 * 64 * 68 * sizeof(float);       // the epilogue's four 64 x 68 tiles (69.6 KB) cover the 4 x 32 x 132 staging tiles (67.6 KB)
  const int64_t total = (int64_t)tm * tn * P;
  hipLaunchKernelGGL(gemm_tn_wide_kernel, dim3((unsigned)(cdiv(total, 8) * 8)), dim3(256), lds, st, g);
  MATCHA_CHECK_LAUNCH("gemm_tn_wide_kernel");
  *P_out = P;
  *slab_stride_out = g.slab_stride;
  return MATCHA_OK;
}

}  // namespace matcha
