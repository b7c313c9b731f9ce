// Per-hyperedge multi-head attention for embed_dim >= 128 (BASELINE configs[3], [4]; Modules.py:417-460 via :561), forward and
// backward, on the ragged token layout -- the same mathematics and lane mapping as attention.hip (one wavefront per
// hyperedge, lane = 8 * head + sub, a lane owns 8 floats of every 64-feature chunk of its head), with the
// loops turned inside out so that the register file holds ONE operand set at a time:
//
//   attention.hip keeps the chunks of EVERY row of two or three operands live at once (Q_i, K_i, dO_i for all i <= L, plus the
//   P, dP matrices): at L = 8 that is 192 + 144 registers -- attn_bwd_kernel<8, 8, 4> spilled 1.3 KB per lane and ran at 1.3 TB/s
//   (BASELINE config 5: 3.5 of 17.9 ms), and from L = 3 up every variant sat at 256 registers = one wave per SIMD on a kernel
//   whose only job is to stream Q/K/V/dO through.
//   Here each phase holds the chunks of ONE operand for all rows (<= 64 registers) and streams the other operand row by row;
//   the probabilities go through 2 KB of LDS per wave (8-lane broadcasts), the padding token's dK / dV partial sums live in
//   LDS (each lane updates only its own addresses), and the kernel runs at two to four waves per SIMD.
//
// The operands are re-read per phase (dO three times, Q / K / V twice) -- from L1 / L2: a hyperedge's rows are touched again
// within microseconds; HBM sees each byte once.
#include "kernels.hpp"

namespace matcha {

namespace {

constexpr int kRows = 8;               // hyperedges per wavefront (amortises the padding-token reduction)

// a lane's 8 consecutive floats of a 64-feature chunk (two 16-byte loads; interleaving the halves so that each load covers whole
// 128-byte lines changed nothing measurable: the kernels are bound by their dependent load -> reduce -> load chains, not by lines)
struct C8 { float v[8]; };
__device__ __forceinline__ C8 ld8(const float* __restrict__ p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  C8 c;
  c.v[0] = a.x; c.v[1] = a.y; c.v[2] = a.z; c.v[3] = a.w; c.v[4] = b.x; c.v[5] = b.y; c.v[6] = b.z; c.v[7] = b.w;
  return c;
}
__device__ __forceinline__ void st8(float* __restrict__ p, const C8& c) {
  *reinterpret_cast<float4*>(p) = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(c.v[4], c.v[5], c.v[6], c.v[7]);
}
__device__ __forceinline__ float dot8(const C8& a, const C8& b) {
  float s = a.v[0] * b.v[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) s += a.v[e] * b.v[e];
  return s;
}
__device__ __forceinline__ void axpy8(C8& y, float w, const C8& x) {
#pragma unroll
  for (int e = 0; e < 8; ++e) y.v[e] += w * x.v[e];
}

template <int kMaxL>
__global__ __launch_bounds__(256, 2) void attn_fwd_wide_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                            const int32_t* __restrict__ row_off, int64_t B, int L, int d, float inv_temp,
                                                            float* __restrict__ O, float* __restrict__ P, int64_t ldk, int hoffk) {
  // ldk / hoffk: row stride and per-head offset of K and V -- (8 d, d) for per-head projections, (d, 0) when every head attends
  // the SAME key / value rows (merged heads: keys = LN_k(x), values = LN_v(x), model.hip)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = lane >> 3, sub = lane & 7;
  const int nchunk = d / 64;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t pad_base = (int64_t)row_off[B] * hd + (int64_t)head * d;
  const int64_t pad_base_k = (int64_t)row_off[B] * ldk + (int64_t)head * hoffk;
  if (blockIdx.x == 0 && wave == 0) {
    // the padding token's query is never evaluated, but its O row is a contraction row of the fc1 weight gradient (times a zero
    // gradient): it must be finite, so zero it (the workspace is not initialised)
    C8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z.v[e] = 0.f;
    for (int c = 0; c < nchunk; ++c) st8(O + pad_base + c * 64 + sub * 8, z);
  }
  for (int it = 0; it < kRows; ++it) {
    const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * kRows + it;
    if (b >= B) return;
    const int t0 = row_off[b];
    const int k = row_off[b + 1] - t0;
    const int n_pad = L - k;
    const float padf = (float)n_pad;
    const int64_t base = (int64_t)t0 * hd + (int64_t)head * d;
    const int64_t base_k = (int64_t)t0 * ldk + (int64_t)head * hoffk;
    float S[kMaxL][kMaxL], Sp[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) {
      Sp[i] = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) S[i][j] = 0.f;
    }
    // scores: K rows held, Q rows streamed
    for (int c = 0; c < nchunk; ++c) {
      const int foff = c * 64 + sub * 8;
      C8 kk[kMaxL], kp;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) kk[j] = ld8(K + base_k + (int64_t)(j < k ? j : 0) * ldk + foff);
      kp = ld8(K + pad_base_k + foff);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          const C8 q = ld8(Q + base + (int64_t)i * hd + foff);
#pragma unroll
          for (int j = 0; j < kMaxL; ++j) S[i][j] += dot8(q, kk[j]);
          Sp[i] += dot8(q, kp);
        }
    }
    // 8-lane reduction, scale, diagonal mask, softmax over the k real slots + n_pad identical padding slots
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
      if (i < k) {
        float mx = -3.4e38f;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j)
          if (j < k) {
            float v = group_sum8_dpp(S[i][j]) * inv_temp;
            if (i == j) v = -1e32f;
            S[i][j] = v;
            mx = fmaxf(mx, v);
          }
        if (n_pad > 0) { Sp[i] = group_sum8_dpp(Sp[i]) * inv_temp; mx = fmaxf(mx, Sp[i]); }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) {
          S[i][j] = (j < k) ? expf(S[i][j] - mx) : 0.f;
          den += S[i][j];
        }
        Sp[i] = (n_pad > 0) ? expf(Sp[i] - mx) : 0.f;
        den += padf * Sp[i];
        const float inv = 1.f / den;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) S[i][j] *= inv;
        Sp[i] *= inv;
      }
    if (P && sub == 0) {          // P[b][head][i][j]: real columns j < k, then the (per-slot) padding probability in column k
      float* pp = P + ((b * MATCHA_N_HEAD + head) * L) * L;
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
#pragma unroll
          for (int j = 0; j < kMaxL; ++j)
            if (j < k) pp[i * L + j] = S[i][j];
          if (n_pad > 0) pp[i * L + k] = Sp[i];
        }
    }
    // O_i = sum_j P_ij V_j + n_pad P_pad_i V_pad: V rows held
    for (int c = 0; c < nchunk; ++c) {
      const int foff = c * 64 + sub * 8;
      C8 v[kMaxL], vp;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) v[j] = ld8(V + base_k + (int64_t)(j < k ? j : 0) * ldk + foff);
      vp = ld8(V + pad_base_k + foff);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          C8 o;
          const float wp = padf * Sp[i];
#pragma unroll
          for (int e = 0; e < 8; ++e) o.v[e] = wp * vp.v[e];
#pragma unroll
          for (int j = 0; j < kMaxL; ++j) axpy8(o, S[i][j], v[j]);     // S[i][j] = 0 for j >= k
          st8(O + base + (int64_t)i * hd + foff, o);
        }
    }
  }
}

// backward: formulas in attention.hip (attn_bwd_kernel).  slab[blk] = {dK_pad [8d], dV_pad [8d]} of the block's hyperedges
// (waves added in a fixed order).
template <int kMaxL, bool kSum>
__global__ __launch_bounds__(256, 2) void attn_bwd_wide_kernel(const float* Q, const float* K, const float* V,
                                                            const float* __restrict__ P, const float* __restrict__ dO,
                                                            const int32_t* __restrict__ row_off, int64_t B, int L, int d, float inv_temp,
                                                            float* dQ, float* dK, float* dV,
                                                            float* __restrict__ slab, int64_t ldk, int hoffk, float* dKs, float* dVs) {
  extern __shared__ float lds[];                 // [4 waves][2][8d] padding-token partial sums | [4 waves][8 heads][kMaxL * kMaxL] probabilities
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = lane >> 3, sub = lane & 7;
  const int nchunk = d / 64;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t pad_base = (int64_t)row_off[B] * hd + (int64_t)head * d;
  const int64_t pad_base_k = (int64_t)row_off[B] * ldk + (int64_t)head * hoffk;     // K / V reads (dK / dV are always written per head)
  (void)pad_base;
  float* padacc = lds + (int64_t)wave * 2 * hd;   // this wave's {dK_pad, dV_pad}; lane (head, sub) owns its 8 features of every chunk
  float* pm = lds + 4 * 2 * hd + (wave * MATCHA_N_HEAD + head) * (kMaxL * kMaxL);
  for (int c = 0; c < nchunk; ++c) {
    const int f = head * d + c * 64 + sub * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { padacc[f + e] = 0.f; padacc[hd + f + e] = 0.f; }
  }
  for (int it = 0; it < kRows; ++it) {
    const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * kRows + it;
    if (b >= B) break;
    const int t0 = row_off[b];
    const int k = row_off[b + 1] - t0;
    const int n_pad = L - k;
    const float padf = (float)n_pad;
    const int64_t base = (int64_t)t0 * hd + (int64_t)head * d;
    const int64_t base_k = (int64_t)t0 * ldk + (int64_t)head * hoffk;
    const float* pp = P + ((b * MATCHA_N_HEAD + head) * L) * L;
    // probabilities of this (hyperedge, head) -> LDS: pm[i * kMaxL + j] = P_ij (0 outside the k x k block); one lane per row
    float Pp[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) Pp[i] = (i < k && n_pad > 0) ? pp[i * L + k] : 0.f;
    if (sub < kMaxL) {
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) pm[sub * kMaxL + j] = (sub < k && j < k) ? pp[sub * L + j] : 0.f;
    }
    // phase A: dP_ij = dO_i . V_j, dPp_i = dO_i . V_pad -- V rows held, dO rows streamed
    float dS[kMaxL][kMaxL], dSp[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) {
      dSp[i] = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) dS[i][j] = 0.f;
    }
    for (int c = 0; c < nchunk; ++c) {
      const int foff = c * 64 + sub * 8;
      C8 v[kMaxL], vp;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) v[j] = ld8(V + base_k + (int64_t)(j < k ? j : 0) * ldk + foff);
      vp = ld8(V + pad_base_k + foff);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          const C8 go = ld8(dO + base + (int64_t)i * hd + foff);
#pragma unroll
          for (int j = 0; j < kMaxL; ++j) dS[i][j] += dot8(go, v[j]);
          dSp[i] += dot8(go, vp);
        }
    }
    // softmax backward: dS_ij = P_ij (dP_ij - sig_i) / temp
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
      if (i < k) {
        float sig = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) { dS[i][j] = group_sum8_dpp(dS[i][j]); sig += pm[i * kMaxL + j] * dS[i][j]; }   // P_ij = 0 for j >= k
        dSp[i] = group_sum8_dpp(dSp[i]);
        sig += padf * Pp[i] * dSp[i];
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) dS[i][j] = pm[i * kMaxL + j] * (dS[i][j] - sig) * inv_temp;
        dSp[i] = Pp[i] * (dSp[i] - sig) * inv_temp;
      } else {
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) dS[i][j] = 0.f;
        dSp[i] = 0.f;
      }
    for (int c = 0; c < nchunk; ++c) {
      const int foff = c * 64 + sub * 8;
      const int f = head * d + foff;
      {   // dQ_i = sum_j dS_ij K_j + n_pad dSp_i K_pad: K rows held
        C8 kk[kMaxL], kp;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) kk[j] = ld8(K + base_k + (int64_t)(j < k ? j : 0) * ldk + foff);
        kp = ld8(K + pad_base_k + foff);
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
          if (i < k) {
            C8 gq;
            const float wp = padf * dSp[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) gq.v[e] = wp * kp.v[e];
#pragma unroll
            for (int j = 0; j < kMaxL; ++j) axpy8(gq, dS[i][j], kk[j]);
            st8(dQ + base + (int64_t)i * hd + foff, gq);
          }
      }
      {   // dK_i = sum_j dS_ji Q_j;  dK_pad += n_pad sum_j dSp_j Q_j: Q rows held
        C8 q[kMaxL];
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) q[j] = ld8(Q + base + (int64_t)(j < k ? j : 0) * hd + foff);
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
          if (i < k) {
            C8 gk;
#pragma unroll
            for (int e = 0; e < 8; ++e) gk.v[e] = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxL; ++j) axpy8(gk, dS[j][i], q[j]);   // dS[j][.] = 0 for j >= k
            if constexpr (kSum) {
              // shared keys: every head's dK_i is a gradient of the SAME row -- the eight lane groups (heads) of the wave are added in a fixed
              // xor order and the row is written once ([T, d]) instead of per head ([T, 8d]) plus a pass that re-reads and sums it
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float v = gk.v[e];
                v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                gk.v[e] = v;
              }
              if (head == 0) st8(dKs + (int64_t)(t0 + i) * d + foff, gk);
            } else {
              st8(dK + base + (int64_t)i * hd + foff, gk);
            }
          }
        if (n_pad > 0) {
          C8 acc = ld8(padacc + f);
#pragma unroll
          for (int j = 0; j < kMaxL; ++j) axpy8(acc, padf * dSp[j], q[j]);
          st8(padacc + f, acc);
        }
      }
      {   // dV_i = sum_j P_ji dO_j;  dV_pad += n_pad sum_j Pp_j dO_j: dO rows held, P columns from LDS (8-lane broadcasts)
        C8 go[kMaxL];
#pragma unroll
        for (int j = 0; j < kMaxL; ++j) go[j] = ld8(dO + base + (int64_t)(j < k ? j : 0) * hd + foff);
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
          if (i < k) {
            C8 gv;
#pragma unroll
            for (int e = 0; e < 8; ++e) gv.v[e] = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxL; ++j) axpy8(gv, pm[j * kMaxL + i], go[j]);
            if constexpr (kSum) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float v = gv.v[e];
                v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                gv.v[e] = v;
              }
              if (head == 0) st8(dVs + (int64_t)(t0 + i) * d + foff, gv);
            } else {
              st8(dV + base + (int64_t)i * hd + foff, gv);
            }
          }
        if (n_pad > 0) {
          C8 acc = ld8(padacc + hd + f);
#pragma unroll
          for (int j = 0; j < kMaxL; ++j) axpy8(acc, padf * Pp[j], go[j]);
          st8(padacc + hd + f, acc);
        }
      }
    }
  }
  // block partial of dK_pad / dV_pad: the four waves' sums added in a fixed order
  __syncthreads();
  float* out = slab + (int64_t)blockIdx.x * 2 * hd;
  for (int i = threadIdx.x; i < 2 * hd; i += 256)
    out[i] = ((lds[i] + lds[2 * hd + i]) + lds[4 * hd + i]) + lds[6 * hd + i];
}

int attn_wide_width(int L) { return L <= 2 ? 2 : (L <= 6 ? L : 8); }

}  // namespace

bool attn_wide_eligible(int d) { return d >= 128 && d % 64 == 0 && !options().disable_wide_gemm; }

int launch_attn_fwd_wide(const float* Q, const float* K, const float* V, const int32_t* row_off, int64_t B, int L, int d, float inv_temp, float* O,
                         float* P, int nblk, hipStream_t st, bool shared_kv) {
  dim3 grid((unsigned)nblk);
  const int64_t ldk = shared_kv ? d : (int64_t)MATCHA_N_HEAD * d;
  const int hoffk = shared_kv ? 0 : d;
#define FWD_W(ML) hipLaunchKernelGGL((attn_fwd_wide_kernel<ML>), grid, dim3(256), 0, st, Q, K, V, row_off, B, L, d, inv_temp, O, P, ldk, hoffk)
  switch (attn_wide_width(L)) {
    case 2: FWD_W(2); break;
    case 3: FWD_W(3); break;
    case 4: FWD_W(4); break;
    case 5: FWD_W(5); break;
    case 6: FWD_W(6); break;
    default: FWD_W(8); break;
  }
#undef FWD_W
  MATCHA_CHECK_LAUNCH("attn_fwd_wide_kernel");
  return MATCHA_OK;
}

int launch_attn_bwd_wide(const float* Q, const float* K, const float* V, const float* P, const float* dO, const int32_t* row_off, int64_t B, int L,
                         int d, float inv_temp, float* dQ, float* dK, float* dV, float* slab, int nblk, hipStream_t st, bool shared_kv, float* dKs,
                         float* dVs) {
  dim3 grid((unsigned)nblk);
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t ldk = shared_kv ? d : hd;
  const int hoffk = shared_kv ? 0 : d;
  const int ml = attn_wide_width(L);
  const size_t lds = ((size_t)4 * 2 * hd + (size_t)4 * MATCHA_N_HEAD * ml * ml) * sizeof(float);
#define BWD_W1(ML, SUM)                                                                                                   \
  do {                                                                                                                    \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_wide_kernel<ML, SUM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((attn_bwd_wide_kernel<ML, SUM>), grid, dim3(256), lds, st, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab, ldk, hoffk, dKs, dVs); \
  } while (0)
#define BWD_W(ML) do { if (dKs) BWD_W1(ML, true); else BWD_W1(ML, false); } while (0)
  switch (ml) {
    case 2: BWD_W(2); break;
    case 3: BWD_W(3); break;
    case 4: BWD_W(4); break;
    case 5: BWD_W(5); break;
    case 6: BWD_W(6); break;
    default: BWD_W(8); break;
  }
#undef BWD_W
#undef BWD_W1
  MATCHA_CHECK_LAUNCH("attn_bwd_wide_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
