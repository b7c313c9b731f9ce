// K9: per-hyperedge multi-head self-attention (Modules.py:417-460 as called from :561), forward + backward, on the
// ragged (CSR) token layout of ragged.hip.
//
// A hyperedge has k <= L <= 8 real nodes, so the whole problem of all 8 heads fits one wavefront:
//   lane = 8*head + sub;  lane (head, sub) owns feature slice [sub*d/8, (sub+1)*d/8) of that head, processed in
//   chunks of CH <= 8 consecutive floats (a chunk of one head = 8 lanes x 32 B = 256 B contiguous, so each token's
//   8d-float Q/K/V row is read as full cache lines); a score is an 8-lane xor-shuffle reduction of partial dots.
//
// Parity with the reference as written (SURVEY.md headline fact 7): only the DIAGONAL is masked (-1e32,
// Modules.py:443-445 with the cached eye-complement :540-556); the key-pad mask never reaches the softmax (call bug
// :612 vs :513), so each of the n_pad = L - k padding slots of the batch-wide layout is an ordinary key/value.  All
// padding slots carry the same K/V row (token index Tr), so their contribution is added in closed form:
//   denominator += n_pad * exp(s_pad_i),   O_i += n_pad * P_pad_i * V_pad,
// and in the backward pass dK_pad / dV_pad are reduced over all hyperedges (fixed-order slabs).
// Queries of padding slots are never needed (their outputs are masked, Modules.py:614, :309).
#include "kernels.hpp"

namespace matcha {

template <int CH>
struct Chunk {
  float v[CH];
};

template <int CH>
__device__ __forceinline__ void load_chunk(const float* __restrict__ p, Chunk<CH>& c) {
  if (CH == 8) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    c.v[0] = a.x; c.v[1] = a.y; c.v[2] = a.z; c.v[3] = a.w; c.v[4 % CH] = b.x; c.v[5 % CH] = b.y; c.v[6 % CH] = b.z; c.v[7 % CH] = b.w;
  } else if (CH == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    c.v[0] = a.x; c.v[1] = a.y; c.v[2 % CH] = a.z; c.v[3 % CH] = a.w;
  } else if (CH == 2) {
    const float2 a = *reinterpret_cast<const float2*>(p);
    c.v[0] = a.x; c.v[1 % CH] = a.y;
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) c.v[i] = p[i];
  }
}
template <int CH>
__device__ __forceinline__ void store_chunk(float* __restrict__ p, const Chunk<CH>& c) {
  if (CH == 8) {
    *reinterpret_cast<float4*>(p) = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(c.v[4 % CH], c.v[5 % CH], c.v[6 % CH], c.v[7 % CH]);
  } else if (CH == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(c.v[0], c.v[1], c.v[2 % CH], c.v[3 % CH]);
  } else if (CH == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(c.v[0], c.v[1 % CH]);
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) p[i] = c.v[i];
  }
}

constexpr int kAttnRowsPerWave = 8;     // hyperedges per wavefront (amortises the dK_pad/dV_pad reduction)

// one wave walks kAttnRowsPerWave hyperedges; 4 waves per 256-thread block
template <int CH, int kMaxL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                       const int32_t* __restrict__ row_off, int64_t B, int L, int d, float inv_temp,
                                                       float* __restrict__ O, float* __restrict__ P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = lane >> 3, sub = lane & 7;
  const int per_lane = d / 8;                    // floats of one head owned by this lane
  const int nchunk = per_lane / CH;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t pad_base = (int64_t)row_off[B] * hd + (int64_t)head * d;
  if (blockIdx.x == 0 && wave == 0) {
    // the padding token's query is never evaluated, but its O row is a contraction row of the fc1 weight gradient
    // (times a zero gradient): it must be finite, so zero it (the workspace is not initialised)
    for (int c = 0; c < nchunk; ++c) {
      Chunk<CH> z;
#pragma unroll
      for (int e = 0; e < CH; ++e) z.v[e] = 0.f;
      store_chunk<CH>(O + pad_base + c * (8 * CH) + sub * CH, z);
    }
  }
  for (int it = 0; it < kAttnRowsPerWave; ++it) {
    const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * kAttnRowsPerWave + it;
    if (b >= B) return;
    const int t0 = row_off[b];
    const int k = row_off[b + 1] - t0;            // real nodes of this hyperedge
    const int n_pad = L - k;                      // padding slots of the batch-wide layout
    const float padf = (float)n_pad;
    const int64_t base = (int64_t)t0 * hd + (int64_t)head * d;

    float S[kMaxL][kMaxL], Sp[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) {
      Sp[i] = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) S[i][j] = 0.f;
    }
    for (int c = 0; c < nchunk; ++c) {
      const int foff = c * (8 * CH) + sub * CH;   // chunk c of this head covers features [c*8*CH, (c+1)*8*CH)
      Chunk<CH> q[kMaxL], kk[kMaxL], kp;
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          load_chunk<CH>(Q + base + i * hd + foff, q[i]);
          load_chunk<CH>(K + base + i * hd + foff, kk[i]);
        }
      if (n_pad > 0) load_chunk<CH>(K + pad_base + foff, kp);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
#pragma unroll
          for (int j = 0; j < kMaxL; ++j)
            if (j < k) {
              float a = 0.f;
#pragma unroll
              for (int e = 0; e < CH; ++e) a += q[i].v[e] * kk[j].v[e];
              S[i][j] += a;
            }
          if (n_pad > 0) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < CH; ++e) a += q[i].v[e] * kp.v[e];
            Sp[i] += a;
          }
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
        for (int j = 0; j < kMaxL; ++j)
          if (j < k) { S[i][j] = expf(S[i][j] - mx); den += S[i][j]; }
        if (n_pad > 0) { Sp[i] = expf(Sp[i] - mx); den += padf * Sp[i]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j)
          if (j < k) S[i][j] *= inv;
        Sp[i] = n_pad > 0 ? Sp[i] * inv : 0.f;
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
    for (int c = 0; c < nchunk; ++c) {
      const int foff = c * (8 * CH) + sub * CH;
      Chunk<CH> v[kMaxL], vp;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (j < k) load_chunk<CH>(V + base + j * hd + foff, v[j]);
      if (n_pad > 0) load_chunk<CH>(V + pad_base + foff, vp);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          Chunk<CH> o;
#pragma unroll
          for (int e = 0; e < CH; ++e) o.v[e] = (n_pad > 0) ? padf * Sp[i] * vp.v[e] : 0.f;
#pragma unroll
          for (int j = 0; j < kMaxL; ++j)
            if (j < k) {
#pragma unroll
              for (int e = 0; e < CH; ++e) o.v[e] += S[i][j] * v[j].v[e];
            }
          store_chunk<CH>(O + base + i * hd + foff, o);
        }
    }
  }
}

// backward (real i, j < k; p = one padding slot, n_pad of them):
//   dV_j = sum_i P_ij dO_i                 dV_pad += n_pad * sum_i Pp_i dO_i
//   dP_ij = dO_i . V_j                     dPp_i = dO_i . V_pad
//   sig_i = sum_j P_ij dP_ij + n_pad Pp_i dPp_i
//   dS_ij = P_ij (dP_ij - sig_i) / temp    dSp_i = Pp_i (dPp_i - sig_i) / temp        (masked diagonal: P_ii = 0)
//   dQ_i = sum_j dS_ij K_j + n_pad dSp_i K_pad ;  dK_j = sum_i dS_ij Q_i ;  dK_pad += n_pad * sum_i dSp_i Q_i
// slab[blk] = {dK_pad [8d], dV_pad [8d]} summed over the block's hyperedges (waves in fixed order).
template <int CH, int kMaxL, int NCHUNK>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* Q, const float* K, const float* V,
                                                       const float* __restrict__ P, const float* __restrict__ dO,
                                                       const int32_t* __restrict__ row_off, int64_t B, int L, int d, float inv_temp,
                                                       float* dQ, float* dK, float* dV,
                                                       float* __restrict__ slab) {
  extern __shared__ float red[];                 // [2][8d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = lane >> 3, sub = lane & 7;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t pad_base = (int64_t)row_off[B] * hd + (int64_t)head * d;
  Chunk<CH> gkp[NCHUNK], gvp[NCHUNK];             // this wave's dK_pad / dV_pad partial (its feature slices)
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
    for (int e = 0; e < CH; ++e) { gkp[c].v[e] = 0.f; gvp[c].v[e] = 0.f; }

  for (int it = 0; it < kAttnRowsPerWave; ++it) {
    const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * kAttnRowsPerWave + it;
    if (b >= B) break;
    const int t0 = row_off[b];
    const int k = row_off[b + 1] - t0;
    const int n_pad = L - k;
    const float padf = (float)n_pad;
    const int64_t base = (int64_t)t0 * hd + (int64_t)head * d;

    float Pm[kMaxL][kMaxL], dP[kMaxL][kMaxL], Pp[kMaxL], dPp[kMaxL];
    const float* pp = P + ((b * MATCHA_N_HEAD + head) * L) * L;
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) {
#pragma unroll
      for (int j = 0; j < kMaxL; ++j) {
        Pm[i][j] = (i < k && j < k) ? pp[i * L + j] : 0.f;
        dP[i][j] = 0.f;
      }
      Pp[i] = (i < k && n_pad > 0) ? pp[i * L + k] : 0.f;
      dPp[i] = 0.f;
    }
    // pass 1: dP_ij = dO_i . V_j ; dPp_i = dO_i . V_pad
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const int foff = c * (8 * CH) + sub * CH;
      Chunk<CH> go[kMaxL], v[kMaxL], vp;
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          load_chunk<CH>(dO + base + i * hd + foff, go[i]);
          load_chunk<CH>(V + base + i * hd + foff, v[i]);
        }
      if (n_pad > 0) load_chunk<CH>(V + pad_base + foff, vp);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
#pragma unroll
          for (int j = 0; j < kMaxL; ++j)
            if (j < k) {
              float a = 0.f;
#pragma unroll
              for (int e = 0; e < CH; ++e) a += go[i].v[e] * v[j].v[e];
              dP[i][j] += a;
            }
          if (n_pad > 0) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < CH; ++e) a += go[i].v[e] * vp.v[e];
            dPp[i] += a;
          }
        }
    }
    // dS (stored in dP / dPp)
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
      if (i < k) {
        float sig = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j)
          if (j < k) { dP[i][j] = group_sum8_dpp(dP[i][j]); sig += Pm[i][j] * dP[i][j]; }
        if (n_pad > 0) { dPp[i] = group_sum8_dpp(dPp[i]); sig += padf * Pp[i] * dPp[i]; }
#pragma unroll
        for (int j = 0; j < kMaxL; ++j)
          if (j < k) dP[i][j] = Pm[i][j] * (dP[i][j] - sig) * inv_temp;
        dPp[i] = (n_pad > 0) ? Pp[i] * (dPp[i] - sig) * inv_temp : 0.f;
      }
    // pass 2
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const int foff = c * (8 * CH) + sub * CH;
      Chunk<CH> q[kMaxL], kk[kMaxL], go[kMaxL], kp;
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          load_chunk<CH>(Q + base + i * hd + foff, q[i]);
          load_chunk<CH>(K + base + i * hd + foff, kk[i]);
          load_chunk<CH>(dO + base + i * hd + foff, go[i]);
        }
      if (n_pad > 0) load_chunk<CH>(K + pad_base + foff, kp);
#pragma unroll
      for (int i = 0; i < kMaxL; ++i)
        if (i < k) {
          Chunk<CH> gq, gk, gv;
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            gq.v[e] = (n_pad > 0) ? padf * dPp[i] * kp.v[e] : 0.f;
            gk.v[e] = 0.f;
            gv.v[e] = 0.f;
          }
#pragma unroll
          for (int j = 0; j < kMaxL; ++j)
            if (j < k) {
#pragma unroll
              for (int e = 0; e < CH; ++e) {
                gq.v[e] += dP[i][j] * kk[j].v[e];      // dQ_i += dS_ij K_j
                gk.v[e] += dP[j][i] * q[j].v[e];       // dK_i += dS_ji Q_j
                gv.v[e] += Pm[j][i] * go[j].v[e];      // dV_i += P_ji dO_j
              }
            }
          store_chunk<CH>(dQ + base + i * hd + foff, gq);
          store_chunk<CH>(dK + base + i * hd + foff, gk);
          store_chunk<CH>(dV + base + i * hd + foff, gv);
          if (n_pad > 0) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
              gkp[c].v[e] += padf * dPp[i] * q[i].v[e];
              gvp[c].v[e] += padf * Pp[i] * go[i].v[e];
            }
          }
        }
    }
  }
  // block partial of dK_pad / dV_pad: waves added in fixed order
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c) {
        const int f = head * d + c * (8 * CH) + sub * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          red[f + e] = (w == 0) ? gkp[c].v[e] : red[f + e] + gkp[c].v[e];
          red[hd + f + e] = (w == 0) ? gvp[c].v[e] : red[hd + f + e] + gvp[c].v[e];
        }
      }
    }
    __syncthreads();
  }
  float* out = slab + (int64_t)blockIdx.x * 2 * hd;
  for (int i = threadIdx.x; i < 2 * hd; i += 256) out[i] = red[i];
}

// dK[Tr] = sum of slabs (dK_pad), dV[Tr] likewise, dQ[Tr] = 0
__global__ __launch_bounds__(1024) void attn_pad_reduce_kernel(const float* __restrict__ slab, int nblk, int64_t hd, const int32_t* __restrict__ row_off,
                                                               int64_t B, float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV) {
  __shared__ float part[16][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;          // index in [0, 2*hd)
  float s0 = 0.f, s1 = 0.f;
  if (i < 2 * hd) {
    int p = q;
    for (; p + 16 < nblk; p += 32) { s0 += slab[(int64_t)p * 2 * hd + i]; s1 += slab[(int64_t)(p + 16) * 2 * hd + i]; }
    if (p < nblk) s0 += slab[(int64_t)p * 2 * hd + i];
  }
  part[q][o] = s0 + s1;
  __syncthreads();
  if (q == 0 && i < 2 * hd) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += part[t][o];
    const int64_t tr = row_off[B];
    if (i < hd) { dK[tr * hd + i] = s; dQ[tr * hd + i] = 0.f; }
    else dV[tr * hd + (i - hd)] = s;
  }
}

// the padding token's row (index *row_dev) of the per-head dK / dV -> its row of the head sums (blockIdx.x = 0: K, 1: V)
__global__ __launch_bounds__(256) void head_sum_row_kernel(const float* __restrict__ dK, const float* __restrict__ dV, const int32_t* __restrict__ row_dev, int d,
                                                           float* __restrict__ dKs, float* __restrict__ dVs) {
  const int64_t tr = *row_dev;
  const float* src = (blockIdx.x == 0 ? dK : dV) + tr * (int64_t)MATCHA_N_HEAD * d;
  float* dst = (blockIdx.x == 0 ? dKs : dVs) + tr * d;
  for (int f = threadIdx.x; f < d; f += 256) {
    float s = src[f];
#pragma unroll
    for (int h = 1; h < MATCHA_N_HEAD; ++h) s += src[(int64_t)h * d + f];
    dst[f] = s;
  }
}

static inline int chunk_of(int d) {
  const int per_lane = d / 8;
  return per_lane >= 8 ? 8 : per_lane;   // 8, 4, 2, 1 (or the exact count when it is 3, 5, 6, 7: handled as CH = 1)
}
// register arrays are sized by the smallest supported width >= L (2,3,4,5,6,8)
static inline int width_of(int L) { return L <= 2 ? 2 : (L <= 6 ? L : 8); }
static inline int attn_blocks(int64_t B) { return (int)cdiv(B, 4 * kAttnRowsPerWave); }

size_t attn_bwd_slab_bytes(int64_t B, int d) { return align_up((size_t)attn_blocks(B) * 2 * MATCHA_N_HEAD * d * sizeof(float), 256); }

#define ATTN_FWD_L(CHV, ...)                                                                                    \
  switch (width_of(L)) {                                                                                        \
    case 2: hipLaunchKernelGGL((attn_fwd_kernel<CHV, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 3: hipLaunchKernelGGL((attn_fwd_kernel<CHV, 3>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 4: hipLaunchKernelGGL((attn_fwd_kernel<CHV, 4>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 5: hipLaunchKernelGGL((attn_fwd_kernel<CHV, 5>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 6: hipLaunchKernelGGL((attn_fwd_kernel<CHV, 6>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    default: hipLaunchKernelGGL((attn_fwd_kernel<CHV, 8>), grid, dim3(256), 0, st, __VA_ARGS__); break;         \
  }

int launch_attn_fwd(const float* Q, const float* K, const float* V, const int32_t* row_off, int64_t B, int L, int d, float* O, float* P,
                    hipStream_t st, bool shared_kv) {
  if (B <= 0) return MATCHA_OK;
  const float inv_temp = 1.0f / sqrtf((float)d);
  dim3 grid((unsigned)attn_blocks(B));
  // algorithmic bytes (upper bound, all L slots real): read Q,K,V, write O (+P)
  ProfScope ps(MATCHA_PROF_ATTN_FWD, 4.0 * ((double)B * L * MATCHA_N_HEAD * d * 4.0 + (double)B * MATCHA_N_HEAD * L * L), st);
  if (attn_wide_eligible(d)) return launch_attn_fwd_wide(Q, K, V, row_off, B, L, d, inv_temp, O, P, attn_blocks(B), st, shared_kv);
  if (shared_kv) { set_error("launch_attn_fwd: shared keys / values need the embed_dim >= 128 kernels"); return MATCHA_EINVAL; }
  switch (chunk_of(d)) {
    case 8: ATTN_FWD_L(8, Q, K, V, row_off, B, L, d, inv_temp, O, P); break;
    case 4: ATTN_FWD_L(4, Q, K, V, row_off, B, L, d, inv_temp, O, P); break;
    case 2: ATTN_FWD_L(2, Q, K, V, row_off, B, L, d, inv_temp, O, P); break;
    default: ATTN_FWD_L(1, Q, K, V, row_off, B, L, d, inv_temp, O, P); break;
  }
  MATCHA_CHECK_LAUNCH("attn_fwd_kernel");
  return MATCHA_OK;
}

#define ATTN_BWD_L(CHV, NCH, ...)                                                                                       \
  switch (width_of(L)) {                                                                                                \
    case 2: hipLaunchKernelGGL((attn_bwd_kernel<CHV, 2, NCH>), grid, dim3(256), lds, st, __VA_ARGS__); break;           \
    case 3: hipLaunchKernelGGL((attn_bwd_kernel<CHV, 3, NCH>), grid, dim3(256), lds, st, __VA_ARGS__); break;           \
    case 4: hipLaunchKernelGGL((attn_bwd_kernel<CHV, 4, NCH>), grid, dim3(256), lds, st, __VA_ARGS__); break;           \
    case 5: hipLaunchKernelGGL((attn_bwd_kernel<CHV, 5, NCH>), grid, dim3(256), lds, st, __VA_ARGS__); break;           \
    case 6: hipLaunchKernelGGL((attn_bwd_kernel<CHV, 6, NCH>), grid, dim3(256), lds, st, __VA_ARGS__); break;           \
    default: hipLaunchKernelGGL((attn_bwd_kernel<CHV, 8, NCH>), grid, dim3(256), lds, st, __VA_ARGS__); break;          \
  }

int launch_attn_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, const int32_t* row_off, int64_t B, int L,
                    int d, float* dQ, float* dK, float* dV, float* slab, hipStream_t st, bool shared_kv, float* dKsum, float* dVsum) {
  if (B <= 0) return MATCHA_OK;
  // dKsum / dVsum (shared keys / values only): [T, d] sums over the heads, written by the kernel itself; dK / dV ([T, 8d]) then only hold
  // the padding token's per-head row (attn_pad_reduce_kernel), which head_sum_row_kernel adds up
  if ((dKsum || dVsum) && !(shared_kv && dKsum && dVsum)) { set_error("launch_attn_bwd: head sums need shared keys / values and both outputs"); return MATCHA_EINVAL; }
  if (shared_kv && !attn_wide_eligible(d)) { set_error("launch_attn_bwd: shared keys / values need the embed_dim >= 128 kernels"); return MATCHA_EINVAL; }
  const float inv_temp = 1.0f / sqrtf((float)d);
  const int nblk = attn_blocks(B);
  dim3 grid((unsigned)nblk);
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const size_t lds = (size_t)2 * hd * sizeof(float);
  {
    // algorithmic bytes (upper bound): read Q,K,V,dO (+P), write dQ,dK,dV
    ProfScope ps(MATCHA_PROF_ATTN_BWD, 4.0 * ((double)B * L * MATCHA_N_HEAD * d * 7.0 + (double)B * MATCHA_N_HEAD * L * L), st);
    const int per_lane = d / 8;
    if (attn_wide_eligible(d)) {
      MATCHA_TRY(launch_attn_bwd_wide(Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab, nblk, st, shared_kv, dKsum, dVsum));
    } else
    switch (chunk_of(d)) {
      case 8:
        if (per_lane == 8) { ATTN_BWD_L(8, 1, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); }
        else if (per_lane == 16) { ATTN_BWD_L(8, 2, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); }
        else if (per_lane == 24) { ATTN_BWD_L(8, 3, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); }
        else { ATTN_BWD_L(8, 4, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); }
        break;
      case 4: ATTN_BWD_L(4, 1, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
      case 2: ATTN_BWD_L(2, 1, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
      default:
        switch (per_lane) {
          case 1: ATTN_BWD_L(1, 1, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
          case 3: ATTN_BWD_L(1, 3, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
          case 5: ATTN_BWD_L(1, 5, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
          case 6: ATTN_BWD_L(1, 6, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
          default: ATTN_BWD_L(1, 7, Q, K, V, P, dO, row_off, B, L, d, inv_temp, dQ, dK, dV, slab); break;
        }
        break;
    }
  }
  MATCHA_CHECK_LAUNCH("attn_bwd_kernel");
  hipLaunchKernelGGL(attn_pad_reduce_kernel, dim3((unsigned)cdiv(2 * hd, 64)), dim3(1024), 0, st, slab, nblk, hd, row_off, B, dQ, dK, dV);
  MATCHA_CHECK_LAUNCH("attn_pad_reduce_kernel");
  if (dKsum) {
    hipLaunchKernelGGL(head_sum_row_kernel, dim3(2), dim3(256), 0, st, dK, dV, row_off + B, d, dKsum, dVsum);
    MATCHA_CHECK_LAUNCH("head_sum_row_kernel");
  }
  return MATCHA_OK;
}

}  // namespace matcha

using namespace matcha;

static int check_attn(int64_t B, int32_t L, int32_t d) {
  MATCHA_CHECK_ARG(L >= 1 && L <= MATCHA_MAX_L, "attention: L=%d outside 1..%d", L, MATCHA_MAX_L);
  MATCHA_CHECK_ARG(d >= 8 && d % 8 == 0 && d <= 256 && ((d / 8) <= 8 ? true : (d / 8) % 8 == 0),
                   "attention: d=%d unsupported (multiple of 8; multiples of 64 above 64)", d);
  MATCHA_CHECK_ARG(B >= 0, "attention: B < 0");
  return MATCHA_OK;
}

extern "C" size_t matcha_attn_bwd_workspace_bytes(int64_t B, int32_t d) { return attn_bwd_slab_bytes(B, d); }

extern "C" int matcha_attn_fwd(const float* Q, const float* K, const float* V, const int32_t* row_off, int64_t B, int32_t L, int32_t d,
                               float* O, float* P, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(Q && K && V && O && row_off, "matcha_attn_fwd: null pointer");
  MATCHA_TRY(check_attn(B, L, d));
  return launch_attn_fwd(Q, K, V, row_off, B, L, d, O, P, (hipStream_t)stream);
}

extern "C" int matcha_attn_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, const int32_t* row_off,
                               int64_t B, int32_t L, int32_t d, float* dQ, float* dK, float* dV, void* ws, size_t ws_bytes,
                               matcha_stream_t stream) {
  MATCHA_CHECK_ARG(Q && K && V && P && dO && dQ && dK && dV && row_off && ws, "matcha_attn_bwd: null pointer");
  MATCHA_TRY(check_attn(B, L, d));
  MATCHA_CHECK_ARG(ws_bytes >= attn_bwd_slab_bytes(B, d), "matcha_attn_bwd: workspace too small");
  return launch_attn_bwd(Q, K, V, P, dO, row_off, B, L, d, dQ, dK, dV, (float*)ws, (hipStream_t)stream);
}
