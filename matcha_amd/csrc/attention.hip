// K9: per-hyperedge multi-head self-attention (Modules.py:417-460 as called from :561), forward + backward.
//
// A hyperedge has L <= 8 slots, so the whole L x L problem of all 8 heads fits one wavefront:
//   lane = 8*head + sub;  lane (head, sub) owns feature slice [sub*d/8, (sub+1)*d/8) of that head,
//   processed in chunks of CH <= 8 consecutive floats (a chunk of one head = 8 lanes x 32 B = 256 B
//   contiguous, so each token's 8d-float Q/K/V row is read as full cache lines);
//   a score is an 8-lane xor-shuffle reduction of per-lane partial dots.
// Parity notes (SURVEY.md headline fact 7): only the DIAGONAL is masked (-1e32, Modules.py:443-445 with the
// cached eye-complement :540-556); the key-pad mask never reaches the softmax (call bug :612 vs :513), so pad
// slots are ordinary keys and values.  The kernel therefore needs no node ids at all.
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

// one wave per hyperedge; 4 hyperedges per 256-thread block
template <int CH, int kMaxL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, int64_t B, int L, int d, float inv_temp,
                                                       float* __restrict__ O, float* __restrict__ P) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int head = lane >> 3, sub = lane & 7;
  const int per_lane = d / 8;                    // floats of one head owned by this lane
  const int nchunk = per_lane / CH;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t base = b * L * hd + (int64_t)head * d;

  float S[kMaxL][kMaxL];
#pragma unroll
  for (int i = 0; i < kMaxL; ++i)
#pragma unroll
    for (int j = 0; j < kMaxL; ++j) S[i][j] = 0.f;

  for (int c = 0; c < nchunk; ++c) {
    const int foff = c * (8 * CH) + sub * CH;   // chunk c of this head covers features [c*8*CH, (c+1)*8*CH)
    Chunk<CH> q[kMaxL], k[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) {
      if (i < L) {
        load_chunk<CH>(Q + base + i * hd + foff, q[i]);
        load_chunk<CH>(K + base + i * hd + foff, k[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (i < L && j < L) {
          float a = 0.f;
#pragma unroll
          for (int e = 0; e < CH; ++e) a += q[i].v[e] * k[j].v[e];
          S[i][j] += a;
        }
  }
  // 8-lane reduction, scale, diagonal mask, softmax over ALL L slots (pads included)
#pragma unroll
  for (int i = 0; i < kMaxL; ++i) {
    if (i < L) {
      float mx = -3.4e38f;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (j < L) {
          float v = group_sum<8>(S[i][j]) * inv_temp;
          if (i == j) v = -1e32f;
          S[i][j] = v;
          mx = fmaxf(mx, v);
        }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (j < L) { S[i][j] = expf(S[i][j] - mx); den += S[i][j]; }
      const float inv = 1.f / den;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (j < L) S[i][j] *= inv;
    }
  }
  if (P && sub == 0) {
    float* pp = P + ((b * MATCHA_N_HEAD + head) * L) * L;
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (i < L && j < L) pp[i * L + j] = S[i][j];
  }
  for (int c = 0; c < nchunk; ++c) {
    const int foff = c * (8 * CH) + sub * CH;
    Chunk<CH> v[kMaxL];
#pragma unroll
    for (int j = 0; j < kMaxL; ++j)
      if (j < L) load_chunk<CH>(V + base + j * hd + foff, v[j]);
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) {
      if (i < L) {
        Chunk<CH> o;
#pragma unroll
        for (int e = 0; e < CH; ++e) o.v[e] = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxL; ++j)
          if (j < L) {
#pragma unroll
            for (int e = 0; e < CH; ++e) o.v[e] += S[i][j] * v[j].v[e];
          }
        store_chunk<CH>(O + base + i * hd + foff, o);
      }
    }
  }
}

// backward: dV_j = sum_i P_ij dO_i ; dP_ij = dO_i . V_j ; dS_ij = P_ij (dP_ij - sum_j' P_ij' dP_ij') / temp ;
//           dQ_i = sum_j dS_ij K_j ; dK_j = sum_i dS_ij Q_i     (masked diagonal: P_ii = 0 -> dS_ii = 0)
template <int CH, int kMaxL>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, const float* __restrict__ P,
                                                       const float* __restrict__ dO, int64_t B, int L, int d, float inv_temp,
                                                       float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int head = lane >> 3, sub = lane & 7;
  const int per_lane = d / 8;
  const int nchunk = per_lane / CH;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const int64_t base = b * L * hd + (int64_t)head * d;

  float Pm[kMaxL][kMaxL], dP[kMaxL][kMaxL];
  const float* pp = P + ((b * MATCHA_N_HEAD + head) * L) * L;
#pragma unroll
  for (int i = 0; i < kMaxL; ++i)
#pragma unroll
    for (int j = 0; j < kMaxL; ++j) {
      Pm[i][j] = (i < L && j < L) ? pp[i * L + j] : 0.f;
      dP[i][j] = 0.f;
    }
  // pass 1: dP_ij = dO_i . V_j
  for (int c = 0; c < nchunk; ++c) {
    const int foff = c * (8 * CH) + sub * CH;
    Chunk<CH> go[kMaxL], v[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
      if (i < L) {
        load_chunk<CH>(dO + base + i * hd + foff, go[i]);
        load_chunk<CH>(V + base + i * hd + foff, v[i]);
      }
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (i < L && j < L) {
          float a = 0.f;
#pragma unroll
          for (int e = 0; e < CH; ++e) a += go[i].v[e] * v[j].v[e];
          dP[i][j] += a;
        }
  }
  // dS (stored in dP)
#pragma unroll
  for (int i = 0; i < kMaxL; ++i)
    if (i < L) {
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (j < L) { dP[i][j] = group_sum<8>(dP[i][j]); dot += Pm[i][j] * dP[i][j]; }
#pragma unroll
      for (int j = 0; j < kMaxL; ++j)
        if (j < L) dP[i][j] = Pm[i][j] * (dP[i][j] - dot) * inv_temp;
    }
  // pass 2
  for (int c = 0; c < nchunk; ++c) {
    const int foff = c * (8 * CH) + sub * CH;
    Chunk<CH> q[kMaxL], k[kMaxL], go[kMaxL];
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
      if (i < L) {
        load_chunk<CH>(Q + base + i * hd + foff, q[i]);
        load_chunk<CH>(K + base + i * hd + foff, k[i]);
        load_chunk<CH>(dO + base + i * hd + foff, go[i]);
      }
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
      if (i < L) {
        Chunk<CH> gq, gk, gv;
#pragma unroll
        for (int e = 0; e < CH; ++e) { gq.v[e] = 0.f; gk.v[e] = 0.f; gv.v[e] = 0.f; }
#pragma unroll
        for (int j = 0; j < kMaxL; ++j)
          if (j < L) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
              gq.v[e] += dP[i][j] * k[j].v[e];      // dQ_i += dS_ij K_j
              gk.v[e] += dP[j][i] * q[j].v[e];      // dK_i += dS_ji Q_j
              gv.v[e] += Pm[j][i] * go[j].v[e];     // dV_i += P_ji dO_j
            }
          }
        store_chunk<CH>(dQ + base + i * hd + foff, gq);
        store_chunk<CH>(dK + base + i * hd + foff, gk);
        store_chunk<CH>(dV + base + i * hd + foff, gv);
      }
  }
}

static inline int chunk_of(int d) {
  const int per_lane = d / 8;
  return per_lane >= 8 ? 8 : per_lane;   // 8, 4, 2, 1
}

// register arrays are sized by the smallest supported width >= L (2,3,4,5,6,8)
static inline int width_of(int L) { return L <= 2 ? 2 : (L <= 6 ? L : 8); }

#define ATTN_DISPATCH_L(KERNEL, CHV, ...)                                                              \
  switch (width_of(L)) {                                                                               \
    case 2: hipLaunchKernelGGL((KERNEL<CHV, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 3: hipLaunchKernelGGL((KERNEL<CHV, 3>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 4: hipLaunchKernelGGL((KERNEL<CHV, 4>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 5: hipLaunchKernelGGL((KERNEL<CHV, 5>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    case 6: hipLaunchKernelGGL((KERNEL<CHV, 6>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
    default: hipLaunchKernelGGL((KERNEL<CHV, 8>), grid, dim3(256), 0, st, __VA_ARGS__); break;         \
  }
#define ATTN_DISPATCH(KERNEL, ...)                                    \
  switch (chunk_of(d)) {                                              \
    case 8: ATTN_DISPATCH_L(KERNEL, 8, __VA_ARGS__); break;           \
    case 4: ATTN_DISPATCH_L(KERNEL, 4, __VA_ARGS__); break;           \
    case 2: ATTN_DISPATCH_L(KERNEL, 2, __VA_ARGS__); break;           \
    default: ATTN_DISPATCH_L(KERNEL, 1, __VA_ARGS__); break;          \
  }

int launch_attn_fwd(const float* Q, const float* K, const float* V, int64_t B, int L, int d, float* O, float* P, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  const float inv_temp = 1.0f / sqrtf((float)d);
  dim3 grid((unsigned)cdiv(B, 4));
  // algorithmic bytes: read Q,K,V, write O (+P)
  ProfScope ps(MATCHA_PROF_ATTN_FWD, 4.0 * ((double)B * L * MATCHA_N_HEAD * d * 4.0 + (double)B * MATCHA_N_HEAD * L * L), st);
  ATTN_DISPATCH(attn_fwd_kernel, Q, K, V, B, L, d, inv_temp, O, P);
  MATCHA_CHECK_LAUNCH("attn_fwd_kernel");
  return MATCHA_OK;
}

int launch_attn_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, int64_t B, int L, int d,
                    float* dQ, float* dK, float* dV, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  const float inv_temp = 1.0f / sqrtf((float)d);
  dim3 grid((unsigned)cdiv(B, 4));
  // algorithmic bytes: read Q,K,V,dO (+P), write dQ,dK,dV
  ProfScope ps(MATCHA_PROF_ATTN_BWD, 4.0 * ((double)B * L * MATCHA_N_HEAD * d * 7.0 + (double)B * MATCHA_N_HEAD * L * L), st);
  ATTN_DISPATCH(attn_bwd_kernel, Q, K, V, P, dO, B, L, d, inv_temp, dQ, dK, dV);
  MATCHA_CHECK_LAUNCH("attn_bwd_kernel");
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

extern "C" int matcha_attn_fwd(const float* Q, const float* K, const float* V, int64_t B, int32_t L, int32_t d, float* O,
                               float* P, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(Q && K && V && O, "matcha_attn_fwd: null pointer");
  MATCHA_TRY(check_attn(B, L, d));
  return launch_attn_fwd(Q, K, V, B, L, d, O, P, (hipStream_t)stream);
}

extern "C" int matcha_attn_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, int64_t B,
                               int32_t L, int32_t d, float* dQ, float* dK, float* dV, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(Q && K && V && P && dO && dQ && dK && dV, "matcha_attn_bwd: null pointer");
  MATCHA_TRY(check_attn(B, L, d));
  return launch_attn_bwd(Q, K, V, P, dO, B, L, d, dQ, dK, dV, (hipStream_t)stream);
}
