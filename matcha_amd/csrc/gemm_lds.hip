// LDS-staged, software-pipelined fp32 MFMA GEMM for the token-major layers (NT: Y = X.W^T; NN: dX = dY.W).
//
// Why LDS here: the MFMA fragment of a row-major operand is "one 16-B piece per lane, consecutive lanes 256 B
// apart", which the texture path serves as 64 separate requests per wave-instruction (measured: the direct-from-
// global version of this kernel sat at ~28 % of the f32 MFMA rate).  Staging makes every global load fully
// coalesced (16 consecutive lanes read one 256-B row segment), lets the four waves of a workgroup share one copy
// of the weight tile, and -- for K <= 64 -- keeps the activation tile resident while the workgroup walks the
// output columns, so X is read from HBM exactly once per layer.
//
//   workgroup = 256 threads = 4 waves; tile = 128 rows x 64 columns x 64 deep; wave = 32 rows x 64 columns
//   a "step" = one (column tile, k chunk): its operands are loaded global->registers WHILE the previous step's
//   64 MFMAs per wave run, written to the other LDS buffer afterwards, one barrier per step
//   LDS: K <= 64: A [128][68] + 2 x B [64][68] = 68 KiB (2 workgroups/CU); K > 64: 2 x A + 2 x B = 102 KiB (1/CU)
//   epilogue flags are COMPILE-TIME (a runtime-flag epilogue costs ~700 basic blocks in the unrolled store loop)
//
// Optional indirections for the adj front end (MultipleEmbedding, Modules.py:176-201), all resolved per workgroup:
//   m_dev      row count read from device memory (segment sizes are only known on the device)
//   a_row_map  gather of A rows, c_row_map scatter of C rows (tokens sorted by chromosome <-> token slots)
//   seg        grouped mode: rows sorted by group, group c uses weight B + c*b_group_stride (per-chromosome W1)
#include "kernels.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 64, kBK = 64, kLd = 68;   // kLd: padded LDS row stride (floats), 16-B aligned rows
// K > 64 walks 32-deep chunks instead (BK = 32, row stride 36): 2 x A + 2 x B = 54 KiB, so two workgroups share a CU -- with
// 64-deep double-buffered chunks (102 KiB) a CU held one workgroup = one wave per SIMD and every barrier and LDS latency showed

int launch_gemm_rm_direct(bool b_kn, const GemmArgs& g, hipStream_t st);   // gemm_f32.hip (unaligned fallback)

template <int FLAGS>
struct Epilogue {
  // FLAGS >= 0: compile-time flag set;  FLAGS < 0: read g.flags at run time (generic path)
  __device__ __forceinline__ static bool has(const GemmArgs& g, int f) { return FLAGS >= 0 ? (FLAGS & f) != 0 : (g.flags & f) != 0; }
};

struct StageRegs {
  float4 a[8];
  float4 b[4];
};
// B tile in LDS: NT: [n = 64][k = BK] (stride BK + 4);  NN (B_KN): [k = BK][n = 64] (stride 68)

template <bool B_KN, int FLAGS, int BK>
__global__ __launch_bounds__(256) void gemm_lds_kernel(GemmArgs g, int n_tiles_per_block, int group_parallel) {
  constexpr int LDK = BK + 4;                        // row stride of the k-major tiles (A, and B in NT mode)
  constexpr int F4 = BK / 4;                         // float4 per k-row segment
  constexpr int RPP = 256 / F4;                      // rows staged per pass (16 or 32)
  constexpr int APASS = kBM / RPP, BPASS = kBN / RPP;
  constexpr int B_SZ = B_KN ? BK * kLd : kBN * LDK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // group_parallel (grouped mode, few row tiles): blockIdx.z = the group this workgroup serves; a row tile that several groups cut is
  // handled by several workgroups side by side instead of one workgroup walking the groups in turn
  const int z = group_parallel ? 0 : (int)blockIdx.z;
  const float* __restrict__ A = g.A[z];
  float* __restrict__ C = g.C[z];
  const int64_t M = g.m_dev ? (int64_t)(*g.m_dev) : g.M;
  const int64_t N = g.N, K = g.K;
  const int64_t m0 = (int64_t)blockIdx.y * kBM;
  if (m0 >= M) return;
  const int64_t m_end = (m0 + kBM < M) ? m0 + kBM : M;
  const int srow = tid / F4, sc4 = (tid % F4) * 4;      // staging of k-major tiles: thread -> (row within a slab, float4 column)
  const int nrow = tid >> 4, nc4 = (tid & 15) * 4;      // staging of the [k][n] tile (NN): 16 k-rows per pass
  const bool kmulti = K > BK;
  float* const As0 = lds;
  float* const As1 = kmulti ? lds + kBM * LDK : lds;
  float* const Bs0 = lds + (kmulti ? 2 : 1) * kBM * LDK;
  float* const Bs1 = Bs0 + B_SZ;

  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (Epilogue<FLAGS>::has(g, MATCHA_EPI_DROPOUT)) {
    key = rng_key(*g.seed, g.stream_id);
    thr = dropout_threshold(g.p_drop);
    keep_scale = 1.f / (1.f - g.p_drop);
  }
  const float* bias = g.bias[z];
  const int nkc = (int)((K + BK - 1) / BK);
  const int64_t tiles_n = (N + kBN - 1) / kBN;
  int n_nt = n_tiles_per_block;
  if ((int64_t)blockIdx.x * n_tiles_per_block + n_nt > tiles_n) n_nt = (int)(tiles_n - (int64_t)blockIdx.x * n_tiles_per_block);
  const int nsteps = n_nt * nkc;

  // A row pointers of this thread's 8 staging rows (constant over steps)
  const float* arow_ptr[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int64_t gm = m0 + srow + RPP * i; gm = gm < M ? gm : M - 1;
    if (g.a_row_map) gm = g.a_row_map[gm];
    arow_ptr[i] = A + gm * g.lda + sc4;
  }

  // groups intersecting this row tile (1 pass when not grouped)
  int c_lo = 0, c_hi = 0;
  if (g.seg) {
    while (c_lo < g.n_groups && g.seg[c_lo + 1] <= m0) ++c_lo;
    c_hi = c_lo;
    while (c_hi < g.n_groups && g.seg[c_hi + 1] < m_end) ++c_hi;
    if (group_parallel) {
      const int gsel = (int)blockIdx.z;
      if (gsel < c_lo || gsel > c_hi) return;            // the whole workgroup: no barrier has been reached
      c_lo = c_hi = gsel;
    }
  }
  bool a_resident = false;
  for (int grp = c_lo; grp <= c_hi; ++grp) {
    int64_t row_lo = m0, row_hi = m_end;
    const float* __restrict__ B = g.B[z];
    if (g.seg) {
      if (grp >= g.n_groups) break;                        // trailing segment (padding slots): no weights, skipped
      const int64_t s0 = g.seg[grp], s1 = g.seg[grp + 1];
      row_lo = s0 > m0 ? s0 : m0;
      row_hi = s1 < m_end ? s1 : m_end;
      if (row_lo >= row_hi) continue;
      B += (int64_t)grp * g.b_group_stride;
    }

    // ---- global -> registers for step s -------------------------------------------------------------------
    auto gload = [&](int s, StageRegs& rg, bool with_a) {
      const int nt_i = s / nkc;
      const int64_t kc = (int64_t)(s - nt_i * nkc) * BK;
      const int64_t n0 = ((int64_t)blockIdx.x * n_tiles_per_block + nt_i) * kBN;
      const bool full_k = kc + BK <= K;
      if (with_a) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
          const float* src = arow_ptr[i] + kc;
          float4 v;
          if (full_k) v = *reinterpret_cast<const float4*>(src);
          else {
            // lda is a multiple of 4 and >= K, so a 16-B load that STARTS below K stays inside the row
            const int64_t k = kc + sc4;
            v = (k < K) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.y = (k + 1 < K) ? v.y : 0.f; v.z = (k + 2 < K) ? v.z : 0.f; v.w = (k + 3 < K) ? v.w : 0.f;
          }
          rg.a[i] = v;
        }
      }
      if (!B_KN) {                                         // B[n][k]: 64 rows x F4 float4
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
          int64_t gn = n0 + srow + RPP * i; gn = gn < N ? gn : N - 1;
          const float* src = B + gn * g.ldb + kc + sc4;
          float4 v;
          if (full_k) v = *reinterpret_cast<const float4*>(src);
          else {
            const int64_t k = kc + sc4;
            v = (k < K) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.y = (k + 1 < K) ? v.y : 0.f; v.z = (k + 2 < K) ? v.z : 0.f; v.w = (k + 3 < K) ? v.w : 0.f;
          }
          rg.b[i] = v;
        }
      } else {                                             // B[k][n]: BK k-rows x 16 float4 along n
        const bool full_n = n0 + kBN <= N;
#pragma unroll
        for (int i = 0; i < BK / 16; ++i) {
          const int64_t gk = kc + nrow + 16 * i;
          const int64_t gkc = gk < K ? gk : K - 1;
          float4 v;
          if (full_n) v = *reinterpret_cast<const float4*>(B + gkc * g.ldb + n0 + nc4);
          else {
            const float* src = B + gkc * g.ldb;
            v.x = (n0 + nc4 + 0 < N) ? src[n0 + nc4 + 0] : 0.f;
            v.y = (n0 + nc4 + 1 < N) ? src[n0 + nc4 + 1] : 0.f;
            v.z = (n0 + nc4 + 2 < N) ? src[n0 + nc4 + 2] : 0.f;
            v.w = (n0 + nc4 + 3 < N) ? src[n0 + nc4 + 3] : 0.f;
          }
          if (gk >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
          rg.b[i] = v;
        }
      }
    };
    auto lstore = [&](const StageRegs& rg, float* As, float* Bs, bool with_a) {
      if (with_a) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) *reinterpret_cast<float4*>(&As[(srow + RPP * i) * LDK + sc4]) = rg.a[i];
      }
      if (!B_KN) {
#pragma unroll
        for (int i = 0; i < BPASS; ++i) *reinterpret_cast<float4*>(&Bs[(srow + RPP * i) * LDK + sc4]) = rg.b[i];
      } else {
#pragma unroll
        for (int i = 0; i < BK / 16; ++i) *reinterpret_cast<float4*>(&Bs[(nrow + 16 * i) * kLd + nc4]) = rg.b[i];
      }
    };

    StageRegs rg;
    {
      const bool wa = kmulti || !a_resident;
      gload(0, rg, wa);
      lstore(rg, As0, Bs0, wa);
      a_resident = true;
    }
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      const bool more = s + 1 < nsteps;
      if (more) gload(s + 1, rg, kmulti);                  // in flight during this step's MFMAs
      const int nt_i = s / nkc;
      const int kci = s - nt_i * nkc;
      if (kci == 0) { acc0 = (f32x16){0}; acc1 = (f32x16){0}; }
      const float* As = (kmulti && buf) ? As1 : As0;
      const float* Bs = buf ? Bs1 : Bs0;
      // ---- 32 rows x 64 columns x BK deep per wave: BK MFMAs ----
      const float* arow = &As[(32 * wave + r) * LDK + 4 * h];
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 8 * c);
        float4 b0, b1;
        if (!B_KN) {
          b0 = *reinterpret_cast<const float4*>(&Bs[r * LDK + 8 * c + 4 * h]);
          b1 = *reinterpret_cast<const float4*>(&Bs[(32 + r) * LDK + 8 * c + 4 * h]);
        } else {
          const float* bp = &Bs[(8 * c + 4 * h) * kLd + r];
          b0 = make_float4(bp[0], bp[kLd], bp[2 * kLd], bp[3 * kLd]);
          b1 = make_float4(bp[32], bp[kLd + 32], bp[2 * kLd + 32], bp[3 * kLd + 32]);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
      }
      if (kci == nkc - 1) {
        // ---- epilogue: bias -> tanh -> +residual -> dropout -> row mask -> *(1 - (aux*aux_scale)^2) -> (+=) ----
        const int64_t n0 = ((int64_t)blockIdx.x * n_tiles_per_block + nt_i) * kBN;
        const int64_t mrow0 = m0 + 32 * wave + 4 * h;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int64_t col = n0 + nt * 32 + r;
          if (col >= N) continue;
          const float bv = Epilogue<FLAGS>::has(g, MATCHA_EPI_BIAS) ? bias[col] : 0.f;
          const f32x16 acc = nt == 0 ? acc0 : acc1;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int64_t lrow = mrow0 + (reg & 3) + 8 * (reg >> 2);
            if (lrow < row_lo || lrow >= row_hi) continue;
            const int64_t row = g.c_row_map ? (int64_t)g.c_row_map[lrow] : lrow;
            float v = acc[reg] + bv;
            const int64_t off = row * g.ldc + col;
            if (Epilogue<FLAGS>::has(g, MATCHA_EPI_TANH)) v = tanhf(v);
            if (Epilogue<FLAGS>::has(g, MATCHA_EPI_RESIDUAL)) v += g.residual[off];
            if (Epilogue<FLAGS>::has(g, MATCHA_EPI_DROPOUT)) {
              const uint32_t crow = g.rng_row_map ? (uint32_t)g.rng_row_map[row] : (uint32_t)row;
              v = (rng_u32(key, crow, (uint32_t)col) >= thr) ? v * keep_scale : 0.f;
            }
            if (Epilogue<FLAGS>::has(g, MATCHA_EPI_ROWMASK)) v = (g.row_ids[row] != 0) ? v : 0.f;
            if (Epilogue<FLAGS>::has(g, MATCHA_EPI_DTANH)) { const float a = g.aux[off] * g.aux_scale; v *= (1.f - a * a); }
            if (Epilogue<FLAGS>::has(g, MATCHA_EPI_ACCUM)) v += C[off];
            C[off] = v;
          }
        }
      }
      if (more) lstore(rg, (kmulti && !buf) ? As1 : As0, buf ? Bs0 : Bs1, kmulti);
      __syncthreads();
    }
  }
}

template <bool B_KN, int FLAGS, int BK>
static void launch_bk(const GemmArgs& g, dim3 grid, int ntpb, hipStream_t st, int gpar) {
  auto kfn = gemm_lds_kernel<B_KN, FLAGS, BK>;
  const size_t b_sz = B_KN ? (size_t)BK * kLd : (size_t)kBN * (BK + 4);
  const size_t lds_bytes = ((size_t)(g.K > BK ? 2 : 1) * kBM * (BK + 4) + 2 * b_sz) * sizeof(float);
  if (lds_bytes > 64 * 1024) {
    static bool configured = false;      // per instantiation
    if (!configured) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      configured = true;
    }
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds_bytes, st, g, ntpb, gpar);
}
template <bool B_KN, int FLAGS>
static void launch_one(const GemmArgs& g, dim3 grid, int ntpb, size_t, hipStream_t st, int gpar = 0) {
  if (g.K > kBK) launch_bk<B_KN, FLAGS, 32>(g, grid, ntpb, st, gpar);      // several k chunks: 32 deep, two workgroups per CU
  else launch_bk<B_KN, FLAGS, 64>(g, grid, ntpb, st, gpar);                // K <= 64: the activation tile stays resident
}

int launch_gemm_rm(bool b_kn, const GemmArgs& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return MATCHA_OK;
  bool vec = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.lda >= g.K || b_kn);
  for (int z = 0; z < g.batch; ++z) vec = vec && (((uintptr_t)g.A[z]) % 16 == 0) && (((uintptr_t)g.B[z]) % 16 == 0);
  if (!b_kn) vec = vec && (g.ldb >= g.K);
  if (b_kn) vec = vec && (g.N % 4 == 0);
  const bool indirect = g.m_dev || g.a_row_map || g.c_row_map || g.seg;
  if (vec && gemm_wide_eligible(b_kn, g)) return launch_gemm_wide(b_kn, g, st);     // embed_dim >= 128: 128 x 128 tiles (gemm_wide.hip)
  if (!vec) {
    if (indirect) { set_error("gemm: row maps / grouped mode need 16-byte aligned operands"); return MATCHA_EINVAL; }
    return launch_gemm_rm_direct(b_kn, g, st);
  }
  const int tiles_n = (int)cdiv(g.N, kBN);
  // K <= 64: one workgroup keeps its activation tile in LDS and walks up to 8 column tiles
  int ntpb = 1;
  if (g.K <= kBK) { ntpb = tiles_n < 8 ? tiles_n : 8; }
  const int64_t tiles_m = cdiv(g.M, kBM);
  // few row tiles (the reference's own 384-row batch: 11): parallelism before reuse -- one column tile per workgroup, and in grouped mode
  // one workgroup per (row tile, group) instead of one per row tile walking its groups (each pass is a global -> LDS -> MFMA -> store
  // round trip of ~10 us)
  if (tiles_m * cdiv(tiles_n, ntpb) < 256) ntpb = 1;
  const int gpar = (g.seg && g.batch == 1 && g.n_groups > 1 && tiles_m * g.n_groups <= 4096) ? 1 : 0;
  const size_t lds_bytes = (size_t)((g.K > kBK ? 2 : 1) * kBM + 2 * kBN) * kLd * sizeof(float);
  dim3 grid((unsigned)cdiv(tiles_n, ntpb), (unsigned)tiles_m, (unsigned)(gpar ? g.n_groups : g.batch));
  ProfScope ps(b_kn ? MATCHA_PROF_GEMM_NN : MATCHA_PROF_GEMM_NT, 2.0 * (double)g.M * (double)g.N * (double)g.K * g.batch, st);
  const int F = g.flags;
  constexpr int B_ = MATCHA_EPI_BIAS, T_ = MATCHA_EPI_TANH, R_ = MATCHA_EPI_RESIDUAL, D_ = MATCHA_EPI_DROPOUT, M_ = MATCHA_EPI_ROWMASK,
                G_ = MATCHA_EPI_DTANH;
  if (!b_kn) {
    switch (F) {
      case 0: launch_one<false, 0>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case B_: launch_one<false, B_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case B_ | T_: launch_one<false, B_ | T_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case B_ | T_ | D_: launch_one<false, B_ | T_ | D_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case B_ | M_: launch_one<false, B_ | M_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case B_ | M_ | D_: launch_one<false, B_ | M_ | D_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case B_ | R_: launch_one<false, B_ | R_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      default: launch_one<false, -1>(g, grid, ntpb, lds_bytes, st, gpar); break;
    }
  } else {
    switch (F) {
      case 0: launch_one<true, 0>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case G_: launch_one<true, G_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case G_ | D_: launch_one<true, G_ | D_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case R_ | M_: launch_one<true, R_ | M_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      case R_ | M_ | D_: launch_one<true, R_ | M_ | D_>(g, grid, ntpb, lds_bytes, st, gpar); break;
      default: launch_one<true, -1>(g, grid, ntpb, lds_bytes, st, gpar); break;
    }
  }
  MATCHA_CHECK_LAUNCH("gemm_lds_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
