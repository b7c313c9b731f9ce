// fp32 GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 64 FLOP/clk/SIMD).
//
// The hyperedge classifier's dense work is token-major: A is [T tokens, K] with K in {d, 8d, n_i}, the
// weight is tiny (<= 8d*d floats) and lives in L2, so neither operand is staged through LDS: every lane
// loads its MFMA fragment straight from global memory (A rows as 16-B vectors, the weight from L2) and the
// matrix pipe -- 64 cycles per instruction -- hides the load latency at 3-4 waves per SIMD.
//
//   NT : C[M,N] = A[M,K] . B[N,K]^T        F.linear / Conv1d(k=1)  (Modules.py:111, :340, :392, :481-483, :495)
//   NN : C[M,N] = A[M,K] . B[K,N]          input gradient of the same layers
//   TN : C[M,N] = A[R,M]^T . B[R,N]        weight gradient; R = tokens, split over workgroups into slabs
//                                          that a second kernel sums in a fixed order (bitwise reproducible)
//
// Fragment maps (cdna_hip_programming.md §3): lane l, r = l & 31, h = l >> 5
//   A operand  A[i = r][k = h]     B operand  B[k = h][j = r]     C/D  col = r, row = (reg&3) + 8*(reg>>2) + 4*h
// A k-group of 8 is consumed as four MFMAs; lane half h supplies k = 4h + c for step c, for BOTH operands,
// so the sum over k is a permutation of the natural order (exact same products, different add order).
#include "kernels.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool VEC>
__device__ __forceinline__ float4 load4_guard(const float* p, int64_t k, int64_t K) {
  // four consecutive k of one row, zero beyond K
  if (VEC) {
    if (k + 3 < K) return *reinterpret_cast<const float4*>(p + k);
    return make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    float4 v;
    v.x = (k + 0 < K) ? p[k + 0] : 0.f;
    v.y = (k + 1 < K) ? p[k + 1] : 0.f;
    v.z = (k + 2 < K) ? p[k + 2] : 0.f;
    v.w = (k + 3 < K) ? p[k + 3] : 0.f;
    return v;
  }
}

// block = 256 threads = 4 waves stacked on M; wave tile = 32 rows x 64 cols; block tile = 128 x 64.
// B_KN = false: B is [N,K] (NT);  true: B is [K,N] (NN).
template <bool B_KN, bool VEC>
__global__ __launch_bounds__(256) void gemm_rm_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int z = blockIdx.z;
  const float* __restrict__ A = g.A[z];
  const float* __restrict__ B = g.B[z];
  float* __restrict__ C = g.C[z];
  const int64_t m0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  if (m0 >= g.M) return;
  const int64_t K = g.K;
  int64_t am = m0 + r; if (am >= g.M) am = g.M - 1;           // clamp: loads stay in bounds, stores are guarded
  const float* arow = A + am * g.lda;
  int64_t bn[2];
  bn[0] = n0 + r;      if (bn[0] >= g.N) bn[0] = g.N - 1;
  bn[1] = n0 + 32 + r; if (bn[1] >= g.N) bn[1] = g.N - 1;
  const bool two = (n0 + 32) < g.N;

  f32x16 acc0 = {0}, acc1 = {0};
  for (int64_t kc = 0; kc < K; kc += 64) {
    float4 af[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) af[c] = load4_guard<VEC>(arow, kc + 8 * c + 4 * h, K);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (nt == 1 && !two) break;
      float4 bf[8];
      if (!B_KN) {
        const float* brow = B + bn[nt] * g.ldb;
#pragma unroll
        for (int c = 0; c < 8; ++c) bf[c] = load4_guard<VEC>(brow, kc + 8 * c + 4 * h, K);
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int64_t k = kc + 8 * c + 4 * h;
          bf[c].x = (k + 0 < K) ? B[(k + 0) * g.ldb + bn[nt]] : 0.f;
          bf[c].y = (k + 1 < K) ? B[(k + 1) * g.ldb + bn[nt]] : 0.f;
          bf[c].z = (k + 2 < K) ? B[(k + 2) * g.ldb + bn[nt]] : 0.f;
          bf[c].w = (k + 3 < K) ? B[(k + 3) * g.ldb + bn[nt]] : 0.f;
        }
      }
      f32x16 acc = nt == 0 ? acc0 : acc1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (kc + 8 * c < K) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].x, bf[c].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].y, bf[c].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].z, bf[c].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].w, bf[c].w, acc, 0, 0, 0);
        }
      }
      if (nt == 0) acc0 = acc; else acc1 = acc;
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const int flags = g.flags;
  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (flags & (MATCHA_EPI_DROPOUT)) {
    key = rng_key(*g.seed, g.stream_id);
    thr = dropout_threshold(g.p_drop);
    keep_scale = 1.f / (1.f - g.p_drop);
  }
  const float* bias = g.bias[z];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    if (nt == 1 && !two) break;
    const int64_t col = n0 + nt * 32 + r;
    if (col >= g.N) continue;
    const float bv = (flags & MATCHA_EPI_BIAS) ? bias[col] : 0.f;
    const f32x16 acc = nt == 0 ? acc0 : acc1;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int64_t row = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row >= g.M) continue;
      float v = acc[reg] + bv;
      const int64_t off = row * g.ldc + col;
      // fixed order: bias -> tanh -> +residual -> dropout -> row mask -> * (1 - (aux*aux_scale)^2)
      if (flags & MATCHA_EPI_TANH) v = tanhf(v);
      if (flags & MATCHA_EPI_RESIDUAL) v += g.residual[off];
      if (flags & MATCHA_EPI_DROPOUT) v = (rng_u32(key, (uint32_t)row, (uint32_t)col) >= thr) ? v * keep_scale : 0.f;
      if (flags & MATCHA_EPI_ROWMASK) v = (g.row_ids[row] != 0) ? v : 0.f;
      if (flags & MATCHA_EPI_DTANH) { const float a = g.aux[off] * g.aux_scale; v *= (1.f - a * a); }
      if (flags & MATCHA_EPI_ACCUM) v += C[off];
      C[off] = v;
    }
  }
}

// ---- TN: weight gradients ---------------------------------------------------------------------
struct GemmTnArgs {
  const float* A;   // [R, M]  (dY)
  const float* B;   // [R, N]  (activations), optionally row-gathered
  const int64_t* b_gather;
  float* slab;      // partition p at slab + p*slab_stride: [M*N]
  float* colslab;   // partition p at colslab + p*slab_stride: [M]; or null
  int64_t slab_stride;
  const int32_t* r_dev;
  int64_t M, N, R, lda, ldb;
  int64_t rows_per_block;
  int tiles_n;
};

// block = 256 threads; one 64x64 output tile x one R-partition; the 4 waves split the partition's rows.
template <bool GATHER>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs g) {
  __shared__ float red[64 * 64];
  __shared__ float redc[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x;
  const int64_t mo0 = (int64_t)(tile / g.tiles_n) * 64;
  const int64_t no0 = (int64_t)(tile % g.tiles_n) * 64;
  const int64_t p = blockIdx.y;
  const int64_t R = g.r_dev ? (int64_t)(*g.r_dev) : g.R;      // row count may live on the device (ragged layout)
  int64_t rpb = g.rows_per_block;
  if (g.r_dev) {                                               // split the ACTUAL rows evenly over the launched partitions
    rpb = (R + gridDim.y - 1) / gridDim.y;
    rpb = (rpb + 31) / 32 * 32;
    if (rpb < 32) rpb = 32;
  }
  const int64_t rbeg = p * rpb;
  int64_t rend = rbeg + rpb; if (rend > R) rend = R;
  if (rend < rbeg) rend = rbeg;                               // empty partition: writes a zero slab
  const bool do_col = (g.colslab != nullptr) && (no0 == 0);

  int64_t am[2], bn[2];
  bool amv[2], bnv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    am[t] = mo0 + 32 * t + r; amv[t] = am[t] < g.M; if (!amv[t]) am[t] = g.M - 1;
    bn[t] = no0 + 32 * t + r; bnv[t] = bn[t] < g.N; if (!bnv[t]) bn[t] = g.N - 1;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
  float cs[2] = {0.f, 0.f};

  // Wave w takes 8-row groups w, w+4, w+8, ... of the partition.  Loads are UNCONDITIONAL (row/column indices
  // clamped into range, out-of-range operands zeroed by a multiplicative 0/1 mask): a predicated load makes hipcc
  // branch around every load and wait vmcnt(0) each time, serialising 16 round trips per group.  The next
  // group's 16 loads are issued before the current group's 16 MFMAs (register double buffer).
  const float amf[2] = {amv[0] ? 1.f : 0.f, amv[1] ? 1.f : 0.f};
  const float bnf[2] = {bnv[0] ? 1.f : 0.f, bnv[1] ? 1.f : 0.f};
  const int64_t last = rend - 1;
  auto load_group = [&](int64_t r0, float (&a)[2][4], float (&b)[2][4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int64_t rr = r0 + 4 * h + c;
      rr = rr < last ? rr : last;
      const int64_t brow = GATHER ? g.b_gather[rr] : rr;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t][c] = g.A[rr * g.lda + am[t]];
        b[t][c] = g.B[brow * g.ldb + bn[t]];
      }
    }
  };
  auto mma_group = [&](int64_t r0, float (&a)[2][4], float (&b)[2][4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float rowf = (r0 + 4 * h + c) < rend ? 1.f : 0.f;
      const float a0 = a[0][c] * (amf[0] * rowf), a1 = a[1][c] * (amf[1] * rowf);
      const float b0 = b[0][c] * bnf[0], b1 = b[1][c] * bnf[1];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      cs[0] += a0;
      cs[1] += a1;
    }
  };
  {
    int64_t r0 = rbeg + 8 * wave;
    float a0[2][4], b0[2][4], a1[2][4], b1[2][4];
    if (r0 < rend) load_group(r0, a0, b0);
    while (r0 < rend) {
      const int64_t rn = r0 + 32;
      if (rn < rend) load_group(rn, a1, b1);
      mma_group(r0, a0, b0);
      r0 = rn;
      if (r0 >= rend) break;
      const int64_t rn2 = r0 + 32;
      if (rn2 < rend) load_group(rn2, a0, b0);
      mma_group(r0, a1, b1);
      r0 = rn2;
    }
  }

  // combine the four waves in a fixed order (0,1,2,3) through LDS -> deterministic
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int col = 32 * j + r;
            const int idx = row * 64 + col;
            red[idx] = (w == 0) ? acc[i][j][reg] : red[idx] + acc[i][j][reg];
          }
      if (do_col) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float s = cs[t] + __shfl_xor(cs[t], 32, kWave);
          if (h == 0) redc[32 * t + r] = (w == 0) ? s : redc[32 * t + r] + s;
        }
      }
    }
    __syncthreads();
  }
  float* slab = g.slab + p * g.slab_stride;
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int64_t row = mo0 + idx / 64, col = no0 + idx % 64;
    if (row < g.M && col < g.N) slab[row * g.N + col] = red[idx];
  }
  if (do_col && threadIdx.x < 64) {
    const int64_t row = mo0 + threadIdx.x;
    if (row < g.M) g.colslab[p * g.slab_stride + row] = redc[threadIdx.x];
  }
}

// out[i] (+)= sum_p slab[p][i] in a FIXED order: block = 64 outputs x 16 partition lanes; lane q sums slabs
// q, q+16, ... (ascending), then the 16 partial sums are added q = 0..15.  Coalesced 256-B reads per slab row.
// Element i < n1 goes to out1[i], element n1 <= i < stride to out2[i - n1] (the column sums behind a TN slab).
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out1, int64_t n1,
                                                           float* __restrict__ out2, int64_t stride, int P, int accumulate,
                                                           const float* __restrict__ scale_dev, float scale) {
  __shared__ float part[16][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;
  const int64_t n = out2 ? stride : n1;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int p = q;
    for (; p + 16 < P; p += 32) { s0 += slab[(int64_t)p * stride + i]; s1 += slab[(int64_t)(p + 16) * stride + i]; }
    if (p < P) s0 += slab[(int64_t)p * stride + i];
  }
  part[q][o] = s0 + s1;
  __syncthreads();
  if (q == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += part[t][o];
    float* dst = i < n1 ? out1 + i : out2 + (i - n1);
    s *= scale_dev ? scale_dev[0] * scale : scale;        // optional output scale (a device scalar times a host scalar)
    *dst = accumulate ? *dst + s : s;
  }
}

int launch_slab_reduce(const float* slab, float* out1, int64_t n1, float* out2, int64_t stride, int P, bool accumulate, hipStream_t st,
                       const float* scale_dev, float scale) {
  const int64_t n = out2 ? stride : n1;
  if (n <= 0) return MATCHA_OK;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, slab, out1, n1, out2, stride, P, accumulate ? 1 : 0,
                     scale_dev, scale);
  MATCHA_CHECK_LAUNCH("slab_reduce_kernel");
  return MATCHA_OK;
}

static void tn_partition(int64_t M, int64_t N, int64_t R, int* tiles_m, int* tiles_n, int* P, int64_t* rows_per_block) {
  *tiles_m = (int)cdiv(M, 64);
  *tiles_n = (int)cdiv(N, 64);
  const int64_t tiles = (int64_t)(*tiles_m) * (*tiles_n);
  int64_t want = cdiv(1024, tiles);                 // ~4 blocks (16 waves) per CU in total
  int64_t maxp = cdiv(R, 64);                       // at least 64 rows (two 8-row groups per wave) per block
  int64_t p = want < maxp ? want : maxp;
  if (p < 1) p = 1;
  int64_t rpb = cdiv(cdiv(R, p), 32) * 32;          // multiple of 32 rows so the waves' 8-row groups tile it
  *rows_per_block = rpb;
  *P = (int)cdiv(R, rpb);
  if (*P < 1) *P = 1;
}

// direct-from-global variant: only for operands that cannot be loaded as aligned 16-B vectors (gemm_lds.hip has the
// LDS-staged kernel every aligned shape uses)
int launch_gemm_rm_direct(bool b_kn, const GemmArgs& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return MATCHA_OK;
  bool vec = (g.K % 4 == 0) && (g.lda % 4 == 0);
  for (int z = 0; z < g.batch; ++z) vec = vec && (((uintptr_t)g.A[z]) % 16 == 0);
  if (!b_kn) {
    vec = vec && (g.ldb % 4 == 0);
    for (int z = 0; z < g.batch; ++z) vec = vec && (((uintptr_t)g.B[z]) % 16 == 0);
  }
  dim3 grid((unsigned)cdiv(g.N, 64), (unsigned)cdiv(g.M, 128), (unsigned)g.batch);
  ProfScope ps(b_kn ? MATCHA_PROF_GEMM_NN : MATCHA_PROF_GEMM_NT, 2.0 * (double)g.M * (double)g.N * (double)g.K * g.batch, st);
  if (b_kn) {
    if (vec) hipLaunchKernelGGL((gemm_rm_kernel<true, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_rm_kernel<true, false>), grid, dim3(256), 0, st, g);
  } else {
    if (vec) hipLaunchKernelGGL((gemm_rm_kernel<false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_rm_kernel<false, false>), grid, dim3(256), 0, st, g);
  }
  MATCHA_CHECK_LAUNCH("gemm_rm_kernel");
  return MATCHA_OK;
}

size_t gemm_tn_ws_bytes(int64_t M, int64_t N, int64_t R) {
  int tm, tn, P; int64_t rpb;
  tn_partition(M, N, R, &tm, &tn, &P, &rpb);
  const size_t a = align_up((size_t)P * (size_t)(M * N + M) * sizeof(float), 256);
  const size_t b = gemm_tn_wide_ws_bytes(M, N, R);           // 0 when the shape is not eligible for the 128 x 128 kernel
  return a > b ? a : b;
}

// C[M,N] (+)= A[R,M]^T . B[R,N];  colsum[M] (+)= sum_r A[r,:]
int launch_gemm_tn(const float* A, const float* B, float* C, float* colsum, int64_t M, int64_t N, int64_t R,
                   int64_t lda, int64_t ldb, const int64_t* b_gather, bool accumulate, void* ws, size_t ws_bytes,
                   hipStream_t st, const int32_t* r_dev, const float* out_scale_dev, float out_scale) {
  if (M <= 0 || N <= 0) return MATCHA_OK;
  if (gemm_tn_wide_eligible(M, N, R, lda, ldb, A, B, b_gather)) {
    int Pw; int64_t stride;
    {
      ProfScope ps(MATCHA_PROF_GEMM_TN, 2.0 * (double)M * (double)N * (double)R, st);
      MATCHA_TRY(launch_gemm_tn_wide(A, B, M, N, R, lda, ldb, colsum != nullptr, ws, ws_bytes, r_dev, &Pw, &stride, st));
    }
    return launch_slab_reduce((const float*)ws, C, M * N, colsum, stride, Pw, accumulate, st, out_scale_dev, out_scale);
  }
  int tm, tn, P; int64_t rpb;
  tn_partition(M, N, R > 0 ? R : 1, &tm, &tn, &P, &rpb);
  const size_t need = (size_t)P * (size_t)(M * N + M) * sizeof(float);
  if (ws_bytes < need) { set_error("gemm TN workspace too small: %zu < %zu", ws_bytes, need); return MATCHA_ENOMEM; }
  GemmTnArgs g;
  g.A = A; g.B = B; g.b_gather = b_gather; g.r_dev = r_dev;
  // one slab row per partition: [M*N outputs | M column sums]
  g.slab = (float*)ws;
  g.slab_stride = M * N + (colsum ? M : 0);
  g.colslab = colsum ? (float*)ws + M * N : nullptr;
  g.M = M; g.N = N; g.R = R; g.lda = lda; g.ldb = ldb; g.rows_per_block = rpb; g.tiles_n = tn;
  {
    ProfScope ps(MATCHA_PROF_GEMM_TN, 2.0 * (double)M * (double)N * (double)R, st);
    if (b_gather) hipLaunchKernelGGL((gemm_tn_kernel<true>), dim3(tm * tn, P), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_tn_kernel<false>), dim3(tm * tn, P), dim3(256), 0, st, g);
  }
  MATCHA_CHECK_LAUNCH("gemm_tn_kernel");
  return launch_slab_reduce(g.slab, C, M * N, colsum, g.slab_stride, P, accumulate, st, out_scale_dev, out_scale);
}

}  // namespace matcha

using namespace matcha;

extern "C" size_t matcha_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t R) { return gemm_tn_ws_bytes(M, N, R); }

extern "C" int matcha_gemm(int32_t op, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                           const matcha_gemm_epilogue* epi, float* colsum, const int64_t* b_row_gather,
                           void* ws, size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(A && B && C, "matcha_gemm: null operand");
  hipStream_t st = (hipStream_t)stream;
  if (op == MATCHA_GEMM_TN) {
    const bool acc = epi && (epi->flags & MATCHA_EPI_ACCUM);
    return launch_gemm_tn(A, B, C, colsum, M, N, K, M, N, b_row_gather, acc, ws, ws_bytes, st);
  }
  MATCHA_CHECK_ARG(op == MATCHA_GEMM_NT || op == MATCHA_GEMM_NN, "matcha_gemm: bad op %d", op);
  GemmArgs g = {};
  g.A[0] = A; g.B[0] = B; g.C[0] = C; g.batch = 1;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = (op == MATCHA_GEMM_NT) ? K : N; g.ldc = N;
  if (epi) {
    g.flags = epi->flags; g.bias[0] = epi->bias; g.residual = epi->residual; g.aux = epi->aux;
    g.row_ids = epi->row_ids; g.seed = epi->seed; g.stream_id = (uint32_t)epi->stream_id; g.p_drop = epi->p_drop; g.aux_scale = epi->aux_scale;
    MATCHA_CHECK_ARG(!(g.flags & MATCHA_EPI_BIAS) || g.bias[0], "matcha_gemm: EPI_BIAS without bias");
    MATCHA_CHECK_ARG(!(g.flags & MATCHA_EPI_DROPOUT) || g.seed, "matcha_gemm: EPI_DROPOUT without seed");
    MATCHA_CHECK_ARG(!(g.flags & MATCHA_EPI_ROWMASK) || g.row_ids, "matcha_gemm: EPI_ROWMASK without row_ids");
    MATCHA_CHECK_ARG(!(g.flags & MATCHA_EPI_RESIDUAL) || g.residual, "matcha_gemm: EPI_RESIDUAL without residual");
    MATCHA_CHECK_ARG(!(g.flags & MATCHA_EPI_DTANH) || g.aux, "matcha_gemm: aux missing");
  }
  return launch_gemm_rm(op == MATCHA_GEMM_NN, g, st);
}
