// Token-level (HBM-bound) kernels of the hyperedge classifier: embedding-row gather + attribute path,
// the shared-statistics LayerNorm in front of Q/K/V, the classifier tail with the per-hyperedge segmented
// reduction + weighted BCE, and their backward passes.
//
// Layout: activations are [T = B*L tokens, d] fp32 row-major; a token row is handled by a group of 16
// adjacent lanes, each owning float4 chunks j = 4*s + 64*c (s = lane & 15), so one wave-instruction reads
// four whole rows (4 x 256 B at d = 64) and every row statistic is a 4-step xor-shuffle reduction.
#include <string.h>

#include "attr_src.hpp"
#include "kernels.hpp"
#include "loss_reduce.hpp"

namespace matcha {

constexpr int kTPT = 16;       // lanes per token row
constexpr int kMaxChunk = 4;   // float4 chunks per lane: d <= 256
constexpr float kLnEps = 1e-5f;

struct Row {
  float4 v[kMaxChunk];
};

template <int NCH>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int s, int d, Row& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = 4 * s + 64 * c;
    r.v[c] = (j < d) ? *reinterpret_cast<const float4*>(p + j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NCH>
__device__ __forceinline__ void store_row(float* __restrict__ p, int s, int d, const Row& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = 4 * s + 64 * c;
    if (j < d) *reinterpret_cast<float4*>(p + j) = r.v[c];
  }
}
// mean / rstd of a row spread over 16 lanes (two-pass, like ATen's LayerNorm on the values themselves)
template <int NCH>
__device__ __forceinline__ void row_stats(const Row& r, int s, int d, float& mean, float& rstd) {
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) sum += (r.v[c].x + r.v[c].y) + (r.v[c].z + r.v[c].w);   // lanes past d hold zeros
  sum = group_sum<kTPT>(sum);
  mean = sum / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (4 * s + 64 * c < d) {
      const float a = r.v[c].x - mean, b = r.v[c].y - mean, e = r.v[c].z - mean, f = r.v[c].w - mean;
      sq += (a * a + b * b) + (e * e + f * f);
    }
  }
  sq = group_sum<kTPT>(sq);
  rstd = 1.0f / sqrtf(sq / (float)d + kLnEps);
}
template <int NCH>
__device__ __forceinline__ void normalize(const Row& in, int s, int d, float mean, float rstd, Row& out) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool ok = 4 * s + 64 * c < d;
    out.v[c].x = ok ? (in.v[c].x - mean) * rstd : 0.f;
    out.v[c].y = ok ? (in.v[c].y - mean) * rstd : 0.f;
    out.v[c].z = ok ? (in.v[c].z - mean) * rstd : 0.f;
    out.v[c].w = ok ? (in.v[c].w - mean) * rstd : 0.f;
  }
}
template <int NCH>
__device__ __forceinline__ void affine(const Row& xh, const Row& g, const Row& b, Row& out) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    out.v[c].x = xh.v[c].x * g.v[c].x + b.v[c].x;
    out.v[c].y = xh.v[c].y * g.v[c].y + b.v[c].y;
    out.v[c].z = xh.v[c].z * g.v[c].z + b.v[c].z;
    out.v[c].w = xh.v[c].w * g.v[c].w + b.v[c].w;
  }
}
template <int NCH>
__device__ __forceinline__ float dot_rows(const Row& a, const Row& b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += (a.v[c].x * b.v[c].x + a.v[c].y * b.v[c].y) + (a.v[c].z * b.v[c].z + a.v[c].w * b.v[c].w);
  return s;
}
template <int NCH>
__device__ __forceinline__ float sum_row(const Row& a) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += (a.v[c].x + a.v[c].y) + (a.v[c].z + a.v[c].w);
  return s;
}
#define ROW_FOREACH(NCH, expr)                                    \
  _Pragma("unroll") for (int c = 0; c < NCH; ++c) {               \
    { auto& X_ = c; (void)X_; }                                   \
    expr                                                          \
  }

// LayerNorm backward for one row given dy*g (=dxh), xhat, rstd: dx = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat))
template <int NCH>
__device__ __forceinline__ void ln_bwd_row(const Row& dxh, const Row& xh, float rstd, int s, int d, Row& dx) {
  float a = group_sum<kTPT>(sum_row<NCH>(dxh)) / (float)d;
  float b = group_sum<kTPT>(dot_rows<NCH>(dxh, xh)) / (float)d;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool ok = 4 * s + 64 * c < d;
    dx.v[c].x = ok ? rstd * (dxh.v[c].x - a - xh.v[c].x * b) : 0.f;
    dx.v[c].y = ok ? rstd * (dxh.v[c].y - a - xh.v[c].y * b) : 0.f;
    dx.v[c].z = ok ? rstd * (dxh.v[c].z - a - xh.v[c].z * b) : 0.f;
    dx.v[c].w = ok ? rstd * (dxh.v[c].w - a - xh.v[c].w * b) : 0.f;
  }
}
template <int NCH>
__device__ __forceinline__ void mul_rows(const Row& a, const Row& b, Row& o) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    o.v[c].x = a.v[c].x * b.v[c].x; o.v[c].y = a.v[c].y * b.v[c].y;
    o.v[c].z = a.v[c].z * b.v[c].z; o.v[c].w = a.v[c].w * b.v[c].w;
  }
}
template <int NCH>
__device__ __forceinline__ void acc_row(Row& a, const Row& b) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) { a.v[c].x += b.v[c].x; a.v[c].y += b.v[c].y; a.v[c].z += b.v[c].z; a.v[c].w += b.v[c].w; }
}
template <int NCH>
__device__ __forceinline__ void acc_mul_row(Row& a, const Row& b, const Row& e) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    a.v[c].x += b.v[c].x * e.v[c].x; a.v[c].y += b.v[c].y * e.v[c].y;
    a.v[c].z += b.v[c].z * e.v[c].z; a.v[c].w += b.v[c].w * e.v[c].w;
  }
}
template <int NCH>
__device__ __forceinline__ void zero_row(Row& a) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) a.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Block-level column sums: every thread holds NV partial rows for its own columns; sum the 16 token slots of
// the 256-thread block (fixed order) and write [NV][d] to this block's slab.
template <int NCH, int NV>
__device__ __forceinline__ void block_colsum_store(Row (&part)[NV], int d, float* __restrict__ slab_blk, float* lds) {
  const int s = threadIdx.x & 15, slot = threadIdx.x >> 4;    // 16 slots
  for (int v = 0; v < NV; ++v) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j = 4 * s + 64 * c;
      if (j < d) *reinterpret_cast<float4*>(lds + slot * 256 + j) = part[v].v[c];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 256) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc += lds[t * 256 + j];
      slab_blk[v * d + j] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K1 + K6 (+ the add at Modules.py:269): x0[t] = rows[t] + attr_table[x[t]] . Wa^T + ba
//   rows[t] = table[x[t]]  (Wrap_Embedding / nn.Embedding, Modules.py:29-34)  or  dense[t] (adj front end)
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ x, int64_t T, int d,
                                                        const float* __restrict__ table, const float* __restrict__ dense,
                                                        AttrSrc attr, int n_attr,
                                                        const float* __restrict__ Wa, const float* __restrict__ ba,
                                                        float* __restrict__ x0, int tok_per_blk, const int32_t* __restrict__ t_dev) {
  if (t_dev) T = *t_dev;                                // ragged layout: the token count lives on the device
  const float* __restrict__ attr_table = attr.table;
  __shared__ int abounds[64];                           // attr_mode 1: chromosome bounds (the attribute row is rebuilt from the node id)
  if (attr.mode == 1 && threadIdx.x < n_attr) abounds[threadIdx.x] = attr.bounds[threadIdx.x];
  // attribute_nn.weight [d, n_attr] is staged TRANSPOSED in LDS ([n_attr][d]) once per workgroup, so that the 16
  // lanes of a token read consecutive float4s (conflict-free; the 16 token groups broadcast) instead of each
  // lane walking a strided column of the weight in global memory.
  extern __shared__ __attribute__((aligned(16))) float wt[];
  for (int idx = threadIdx.x; idx < n_attr * d; idx += 256) {
    const int j = idx / n_attr, c0 = idx - j * n_attr;
    wt[c0 * d + j] = Wa[idx];
  }
  __syncthreads();
  const int s = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int grp = (threadIdx.x & 63) & ~15;            // first lane of this token's 16-lane group inside the wave
  Row bias;
  load_row<NCH>(ba, s, d, bias);
  const int64_t t0 = (int64_t)blockIdx.x * tok_per_blk;
  // TWO tokens per trip and every load of a trip requested before the first use: both ids (already fetched during the previous trip),
  // both embedding rows and ALL attribute values of both tokens (n_attr <= 32: two 16-value pieces each; fetched piece by piece
  // inside the product loop each piece cost another HBM round trip).  A table that lives in HBM answers in ~2 us: one row per group
  // at a time with the attribute pieces in sequence read the 1 GiB C5 table at 19 % of the HBM roof.
  int64_t ida = 0, idb = 0;
  {
    const int64_t ta = t0 + slot, tb = ta + 16;
    ida = (slot < tok_per_blk && ta < T) ? x[ta] : 0;
    idb = (slot + 16 < tok_per_blk && tb < T) ? x[tb] : 0;
  }
  for (int i = slot; i < tok_per_blk; i += 32) {
    const int64_t ta = t0 + i, tb = ta + 16;
    if (ta >= T) break;
    const bool hb = (i + 16 < tok_per_blk) && tb < T;
    const int64_t ca = ida, cb = idb;
    {                                                   // ids of the next trip
      const int64_t na = ta + 32, nb = tb + 32;
      ida = (i + 32 < tok_per_blk && na < T) ? x[na] : 0;
      idb = (i + 48 < tok_per_blk && nb < T) ? x[nb] : 0;
    }
    Row ea, eb;
    if (table) { load_row<NCH>(table + ca * d, s, d, ea); load_row<NCH>(table + cb * d, s, d, eb); }
    else if (dense) { load_row<NCH>(dense + ta * d, s, d, ea); load_row<NCH>(dense + (hb ? tb : ta) * d, s, d, eb); }
    else { zero_row<NCH>(ea); zero_row<NCH>(eb); }
    Row oa = bias, ob = bias;
    if (attr.mode == 1) {
      // one-hot chromosome || coordinate: the product has two non-zero terms, W[:, chromosome] and coordinate * W[:, n_attr - 1], added in
      // column order like the general loop below (whose other terms are exact zeros): bit-identical to it, and no second random row
      int cla, clb; float cda, cdb;
      attr_decode(attr, abounds, (int)ca, cla, cda);
      attr_decode(attr, abounds, (int)cb, clb, cdb);
      const float ona = cla >= 0 ? 1.f : 0.f, onb = clb >= 0 ? 1.f : 0.f;
      const float* wca = wt + (cla >= 0 ? cla : 0) * d;
      const float* wcb = wt + (clb >= 0 ? clb : 0) * d;
      const float* wl = wt + (n_attr - 1) * d;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int j = 4 * s + 64 * c;
        if (j < d) {
          const float4 wa4 = *reinterpret_cast<const float4*>(wca + j), wb4 = *reinterpret_cast<const float4*>(wcb + j);
          const float4 wl4 = *reinterpret_cast<const float4*>(wl + j);
          oa.v[c].x += ona * wa4.x; oa.v[c].y += ona * wa4.y; oa.v[c].z += ona * wa4.z; oa.v[c].w += ona * wa4.w;
          ob.v[c].x += onb * wb4.x; ob.v[c].y += onb * wb4.y; ob.v[c].z += onb * wb4.z; ob.v[c].w += onb * wb4.w;
          oa.v[c].x += cda * wl4.x; oa.v[c].y += cda * wl4.y; oa.v[c].z += cda * wl4.z; oa.v[c].w += cda * wl4.w;
          ob.v[c].x += cdb * wl4.x; ob.v[c].y += cdb * wl4.y; ob.v[c].z += cdb * wl4.z; ob.v[c].w += cdb * wl4.w;
        }
      }
    } else {
    const float* arow_a = attr_table + ca * attr.ld;
    const float* arow_b = attr_table + cb * attr.ld;
    float ava[2], avb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ava[q] = (16 * q + s < n_attr) ? arow_a[16 * q + s] : 0.f;      // 16 attribute values per coalesced load
      avb[q] = (16 * q + s < n_attr) ? arow_b[16 * q + s] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int base = 16 * q;
      const int nc = n_attr - base < 16 ? n_attr - base : 16;
      for (int c0 = 0; c0 < nc; ++c0) {
        const float a = __shfl(ava[q], grp + c0, kWave), b = __shfl(avb[q], grp + c0, kWave);
        const float* wrow = wt + (base + c0) * d;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int j = 4 * s + 64 * c;
          if (j < d) {
            const float4 wv = *reinterpret_cast<const float4*>(wrow + j);
            oa.v[c].x += a * wv.x; oa.v[c].y += a * wv.y; oa.v[c].z += a * wv.z; oa.v[c].w += a * wv.w;
            ob.v[c].x += b * wv.x; ob.v[c].y += b * wv.y; ob.v[c].z += b * wv.z; ob.v[c].w += b * wv.w;
          }
        }
      }
    }
    for (int base = 32; base < n_attr; base += 16) {    // n_attr > 32 (not a MATCHA shape): the remaining pieces in sequence
      const float av0 = (base + s < n_attr) ? arow_a[base + s] : 0.f, av1 = (base + s < n_attr) ? arow_b[base + s] : 0.f;
      const int nc = n_attr - base < 16 ? n_attr - base : 16;
      for (int c0 = 0; c0 < nc; ++c0) {
        const float a = __shfl(av0, grp + c0, kWave), b = __shfl(av1, grp + c0, kWave);
        const float* wrow = wt + (base + c0) * d;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int j = 4 * s + 64 * c;
          if (j < d) {
            const float4 wv = *reinterpret_cast<const float4*>(wrow + j);
            oa.v[c].x += a * wv.x; oa.v[c].y += a * wv.y; oa.v[c].z += a * wv.z; oa.v[c].w += a * wv.w;
            ob.v[c].x += b * wv.x; ob.v[c].y += b * wv.y; ob.v[c].z += b * wv.z; ob.v[c].w += b * wv.w;
          }
        }
      }
    }
    }
    acc_row<NCH>(oa, ea);
    store_row<NCH>(x0 + ta * d, s, d, oa);
    if (hb) {
      acc_row<NCH>(ob, eb);
      store_row<NCH>(x0 + tb * d, s, d, ob);
    }
  }
}

// K1 alone: rows[t] = table[ids[t]]   (Wrap_Embedding.forward, Modules.py:33-34; save_embeddings main.py:471)
// A 16-lane group gathers TWO rows per trip (both rows' loads are issued before the first store: twice the bytes in
// flight per wave) and writes them with non-temporal stores -- the output is read by somebody else much later, and plain
// stores made the write-back compete with the table rows for L2 (tools/ubench/gather_variants.hip: 33.5 -> 37.6 % of the
// HBM-read roof on a 4 GB table, 39.1 -> 42.7 % on a 256 MB one).  A materialising gather moves as many bytes out as in,
// so it is bound by the ~6.3 TB/s read+write copy ceiling, i.e. ~39 % of the 8 TB/s read roof.
// Ids outside [0, n_nodes] are flagged in `status` and read as row 0 (the reference raises IndexError, Modules.py:34).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, int64_t T, int d,
                                                          const float* __restrict__ table, int64_t n_nodes, float* __restrict__ rows,
                                                          int32_t* __restrict__ status) {
  const int s = threadIdx.x & 15;
  const int64_t t0 = ((int64_t)blockIdx.x * 16 + (threadIdx.x >> 4)) * 2;
  if (t0 >= T) return;
  const bool two = t0 + 1 < T;
  int64_t id0 = ids[t0], id1 = two ? ids[t0 + 1] : 0;
  if (id0 < 0 || id0 > n_nodes || id1 < 0 || id1 > n_nodes) {
    if (status) atomicOr(status, MATCHA_STATUS_BAD_ID);
    if (id0 < 0 || id0 > n_nodes) id0 = 0;
    if (id1 < 0 || id1 > n_nodes) id1 = 0;
  }
  Row e0, e1;
  load_row<NCH>(table + id0 * d, s, d, e0);
  load_row<NCH>(table + id1 * d, s, d, e1);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = 4 * s + 64 * c;
    if (j < d) {
      const f32x4_t v0 = {e0.v[c].x, e0.v[c].y, e0.v[c].z, e0.v[c].w};
      __builtin_nontemporal_store(v0, reinterpret_cast<f32x4_t*>(rows + t0 * d + j));
      if (two) {
        const f32x4_t v1 = {e1.v[c].x, e1.v[c].y, e1.v[c].z, e1.v[c].w};
        __builtin_nontemporal_store(v1, reinterpret_cast<f32x4_t*>(rows + (t0 + 1) * d + j));
      }
    }
  }
}

// flag ids outside [0, n_nodes] (entry points that hand raw ids to kernels which bucket them instead of indexing a table)
__global__ __launch_bounds__(256) void check_ids_kernel(const int64_t* __restrict__ ids, int64_t T, int64_t n_nodes, int32_t* __restrict__ status) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < T) {
    const int64_t id = ids[t];
    if (id < 0 || id > n_nodes) atomicOr(status, MATCHA_STATUS_BAD_ID);
  }
}

// Classifier.get_embedding's outputs in the reference's padded [B, L, d] layout (Modules.py:261-276, :611-617) from the ragged
// activations: dynamic[b,l] = LayerNorm_pff(H2[t]) for a real slot (token t), 0 for a padding slot (the * non_pad_mask of :614);
// static[b,l] = X[t], or the shared padding token's X for a padding slot.  16 lanes per slot.
template <int NCH>
__global__ __launch_bounds__(256) void expand_embedding_kernel(const int64_t* __restrict__ x, int64_t B, int L, int d,
                                                               const int32_t* __restrict__ row_off, const float* __restrict__ H2,
                                                               const float* __restrict__ X, const float* __restrict__ gp,
                                                               const float* __restrict__ bp, float* __restrict__ dynamic,
                                                               float* __restrict__ static_) {
  const int s = threadIdx.x & 15;
  const int64_t slot = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (slot >= B * L) return;
  const int64_t b = slot / L;
  const int l = (int)(slot - b * L);
  int nth = 0;
  for (int i = 0; i < l; ++i) nth += x[b * L + i] != 0 ? 1 : 0;
  const bool real = x[slot] != 0;
  const int64_t t = real ? (int64_t)row_off[b] + nth : (int64_t)row_off[B];      // row_off[B] = the shared padding token
  Row xs, h, hh, u;
  load_row<NCH>(X + t * d, s, d, xs);
  store_row<NCH>(static_ + slot * d, s, d, xs);
  if (real) {
    Row G, Bb;
    load_row<NCH>(gp, s, d, G); load_row<NCH>(bp, s, d, Bb);
    float m, r;
    load_row<NCH>(H2 + t * d, s, d, h);
    row_stats<NCH>(h, s, d, m, r); normalize<NCH>(h, s, d, m, r, hh); affine<NCH>(hh, G, Bb, u);
  } else {
    zero_row<NCH>(u);
  }
  store_row<NCH>(dynamic + slot * d, s, d, u);
}

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// dtable[x[t]] += dx0[t]  (row 0 = padding_idx never receives a gradient)
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ x, int64_t T, int d,
                                                            const float* __restrict__ dx0, float* __restrict__ dtable,
                                                            const int32_t* __restrict__ t_dev) {
  if (t_dev) T = *t_dev;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * d) return;
  const int64_t t = i / d;
  const int j = (int)(i - t * d);
  const int64_t id = x[t];
  if (id != 0) atomicAdd(dtable + id * d + j, dx0[i]);
}

// ------------------------------------------------------------------------------------------------
// K8, LayerNorm part (Modules.py:519-521): the three LayerNorms normalise the SAME row, so x-hat and the
// statistics are computed once and three affine variants are written.
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void ln3_fwd_kernel(const float* __restrict__ X, int64_t T, int d,
                                                      const float* gq, const float* bq, const float* gk, const float* bk,
                                                      const float* gv, const float* bv, float* __restrict__ qin,
                                                      float* __restrict__ kin, float* __restrict__ vin, float* __restrict__ stats,
                                                      const int32_t* __restrict__ t_dev) {
  if (t_dev) T = *t_dev;
  const int s = threadIdx.x & 15;
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (t >= T) return;
  Row x, xh, g, b, o;
  load_row<NCH>(X + t * d, s, d, x);
  float mean, rstd;
  row_stats<NCH>(x, s, d, mean, rstd);
  normalize<NCH>(x, s, d, mean, rstd, xh);
  load_row<NCH>(gq, s, d, g); load_row<NCH>(bq, s, d, b); affine<NCH>(xh, g, b, o); store_row<NCH>(qin + t * d, s, d, o);
  load_row<NCH>(gk, s, d, g); load_row<NCH>(bk, s, d, b); affine<NCH>(xh, g, b, o); store_row<NCH>(kin + t * d, s, d, o);
  load_row<NCH>(gv, s, d, g); load_row<NCH>(bv, s, d, b); affine<NCH>(xh, g, b, o); store_row<NCH>(vin + t * d, s, d, o);
  if (s == 0 && stats) { stats[2 * t] = mean; stats[2 * t + 1] = rstd; }
}

// Backward of the three LayerNorms + the static branch's gradient + tanh' of Modules.py:270:
//   dZ0 = ( LNbwd(dqin*gq + dkin*gk + dvin*gv) + dXs ) * (1 - X^2)
// and the six LayerNorm parameter gradients as per-block column sums: slab[blk][6][d] =
//   {dgq, dbq, dgk, dbk, dgv, dbv}.
template <int NCH>
__global__ __launch_bounds__(256) void ln3_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dqin,
                                                      const float* __restrict__ dkin, const float* __restrict__ dvin,
                                                      const float* __restrict__ dXs, int64_t T, int d,
                                                      const float* gq, const float* gk, const float* gv,
                                                      float* __restrict__ dZ0, float* __restrict__ slab, int tok_per_blk,
                                                      const int32_t* __restrict__ t_dev) {
  __shared__ float lds[16 * 256];
  if (t_dev) {                                          // split the ACTUAL tokens evenly over the launched workgroups
    T = *t_dev;
    tok_per_blk = (int)((T + gridDim.x - 1) / gridDim.x);
    tok_per_blk = (tok_per_blk + 15) / 16 * 16;
  }
  const int s = threadIdx.x & 15, slot = threadIdx.x >> 4;
  Row Gq, Gk, Gv;
  load_row<NCH>(gq, s, d, Gq); load_row<NCH>(gk, s, d, Gk); load_row<NCH>(gv, s, d, Gv);
  Row part[6];
#pragma unroll
  for (int v = 0; v < 6; ++v) zero_row<NCH>(part[v]);
  const int64_t t0 = (int64_t)blockIdx.x * tok_per_blk;
  for (int i = slot; i < tok_per_blk; i += 16) {
    const int64_t t = t0 + i;
    if (t >= T) break;
    Row x, xh, dq, dk, dv, dxh, dx, ds;
    load_row<NCH>(X + t * d, s, d, x);
    float mean, rstd;
    row_stats<NCH>(x, s, d, mean, rstd);
    normalize<NCH>(x, s, d, mean, rstd, xh);
    load_row<NCH>(dqin + t * d, s, d, dq);
    load_row<NCH>(dkin + t * d, s, d, dk);
    load_row<NCH>(dvin + t * d, s, d, dv);
    acc_mul_row<NCH>(part[0], dq, xh); acc_row<NCH>(part[1], dq);
    acc_mul_row<NCH>(part[2], dk, xh); acc_row<NCH>(part[3], dk);
    acc_mul_row<NCH>(part[4], dv, xh); acc_row<NCH>(part[5], dv);
    mul_rows<NCH>(dq, Gq, dxh);
    acc_mul_row<NCH>(dxh, dk, Gk);
    acc_mul_row<NCH>(dxh, dv, Gv);
    ln_bwd_row<NCH>(dxh, xh, rstd, s, d, dx);
    load_row<NCH>(dXs + t * d, s, d, ds);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      dx.v[c].x = (dx.v[c].x + ds.v[c].x) * (1.f - x.v[c].x * x.v[c].x);
      dx.v[c].y = (dx.v[c].y + ds.v[c].y) * (1.f - x.v[c].y * x.v[c].y);
      dx.v[c].z = (dx.v[c].z + ds.v[c].z) * (1.f - x.v[c].z * x.v[c].z);
      dx.v[c].w = (dx.v[c].w + ds.v[c].w) * (1.f - x.v[c].w * x.v[c].w);
    }
    store_row<NCH>(dZ0 + t * d, s, d, dx);
  }
  block_colsum_store<NCH, 6>(part, d, slab + (int64_t)blockIdx.x * 6 * d, lds);
}

// ------------------------------------------------------------------------------------------------
// K12 + K13: classifier tail (Modules.py:290-311) and weighted BCE-with-logits (main.py:56).
// One 16-lane group walks the L tokens of ONE hyperedge, so the masked mean over the k-mer is a
// register-resident segmented reduction.  Pad tokens (x == 0) contribute nothing (non_pad_mask, :309).
//   u   = LN_pff(H2) * mask          (pff_n1's LayerNorm, Modules.py:373-374, masked at :614)
//   dn  = LN1(u), sn = LN2(X)        (:290-291)
//   out = sum_j (dn_j - sn_j)^2 * wc_j + bc        (:295-299)
//   logit = sum_t out_t / (k + 1e-15)              (:309-311)
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void head_fwd_kernel(const int32_t* __restrict__ row_off, const float* __restrict__ H2,
                                                       const float* __restrict__ X, int64_t B, int L, int d, HeadParams hp,
                                                       const float* __restrict__ y, const float* __restrict__ w,
                                                       float* __restrict__ logits, float* __restrict__ row_loss) {
  const int s = threadIdx.x & 15;
  const int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (b >= B) return;
  Row Gp, Bp, G1, B1, G2, B2, Wc;
  load_row<NCH>(hp.gp, s, d, Gp); load_row<NCH>(hp.bp, s, d, Bp);
  load_row<NCH>(hp.g1, s, d, G1); load_row<NCH>(hp.b1, s, d, B1);
  load_row<NCH>(hp.g2, s, d, G2); load_row<NCH>(hp.b2, s, d, B2);
  load_row<NCH>(hp.wc, s, d, Wc);
  const float bc = hp.bc[0];
  float total = 0.f, cnt = 0.f;
  const int t_lo = row_off[b], t_hi = row_off[b + 1];   // the hyperedge's real tokens (ragged layout: pads are not stored)
  for (int64_t t = t_lo; t < t_hi; ++t) {
    Row h, hh, u, uh, dn, xr, xh, sn;
    float m, r;
    load_row<NCH>(H2 + t * d, s, d, h);
    row_stats<NCH>(h, s, d, m, r); normalize<NCH>(h, s, d, m, r, hh); affine<NCH>(hh, Gp, Bp, u);
    row_stats<NCH>(u, s, d, m, r); normalize<NCH>(u, s, d, m, r, uh); affine<NCH>(uh, G1, B1, dn);
    load_row<NCH>(X + t * d, s, d, xr);
    row_stats<NCH>(xr, s, d, m, r); normalize<NCH>(xr, s, d, m, r, xh); affine<NCH>(xh, G2, B2, sn);
    float o = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float a = dn.v[c].x - sn.v[c].x, e = dn.v[c].y - sn.v[c].y, f = dn.v[c].z - sn.v[c].z, g = dn.v[c].w - sn.v[c].w;
      o += (a * a * Wc.v[c].x + e * e * Wc.v[c].y) + (f * f * Wc.v[c].z + g * g * Wc.v[c].w);
    }
    o = group_sum<kTPT>(o) + bc;
    total += o;
    cnt += 1.f;
  }
  const float z = total / (cnt + 1e-15f);
  if (s == 0) {
    logits[b] = z;
    if (row_loss) {
      // binary_cross_entropy_with_logits: w * (max(z,0) - z*y + log1p(exp(-|z|)))
      const float yy = y[b];
      row_loss[b] = w[b] * (fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z))));
    }
  }
}

// Backward of the tail.  dlogit[b] is either given (autograd glue) or derived from the BCE:
//   dlogit = alpha * w * (sigmoid(z) - y) / B.
// Writes dH2 [T,d] (gradient w.r.t. pff_n1's pre-LayerNorm sum) and dXs [T,d] (gradient into X through
// the static branch); parameter gradients as per-block column sums slab[blk][7][d] + slab_bc[blk]:
//   {dgp, dbp, dg1, db1, dg2, db2, dwc}, dbc.
template <int NCH>
__global__ __launch_bounds__(256) void head_bwd_kernel(const int32_t* __restrict__ row_off, const float* __restrict__ H2,
                                                       const float* __restrict__ X, int64_t B, int L, int d, HeadParams hp,
                                                       const float* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ logits, const float* __restrict__ dlogits,
                                                       float alpha, float* __restrict__ dH2, float* __restrict__ dXs,
                                                       float* __restrict__ slab, int rows_per_blk) {
  __shared__ float lds[16 * 256];
  __shared__ float lds_bc[16];
  const int s = threadIdx.x & 15, slot = threadIdx.x >> 4;
  Row Gp, Bp, G1, B1, G2, B2, Wc;
  load_row<NCH>(hp.gp, s, d, Gp); load_row<NCH>(hp.bp, s, d, Bp);
  load_row<NCH>(hp.g1, s, d, G1); load_row<NCH>(hp.b1, s, d, B1);
  load_row<NCH>(hp.g2, s, d, G2); load_row<NCH>(hp.b2, s, d, B2);
  load_row<NCH>(hp.wc, s, d, Wc);
  Row part[7];
#pragma unroll
  for (int v = 0; v < 7; ++v) zero_row<NCH>(part[v]);
  float part_bc = 0.f;
  const int64_t b0 = (int64_t)blockIdx.x * rows_per_blk;
  for (int i = slot; i < rows_per_blk; i += 16) {
    const int64_t b = b0 + i;
    if (b >= B) break;
    const int t_lo = row_off[b], t_hi = row_off[b + 1];
    const float cnt = (float)(t_hi - t_lo);
    float dz;
    if (dlogits) dz = dlogits[b];
    else {
      const float z = logits[b];
      const float sg = 1.f / (1.f + expf(-z));
      dz = alpha * w[b] * (sg - y[b]) / (float)B;
    }
    const float dout = dz / (cnt + 1e-15f);
    for (int64_t t = t_lo; t < t_hi; ++t) {
      Row h, hh, u, uh, dn, xr, xh, sn;
      float m, rh, ru, rx;
      load_row<NCH>(H2 + t * d, s, d, h);
      row_stats<NCH>(h, s, d, m, rh); normalize<NCH>(h, s, d, m, rh, hh); affine<NCH>(hh, Gp, Bp, u);
      row_stats<NCH>(u, s, d, m, ru); normalize<NCH>(u, s, d, m, ru, uh); affine<NCH>(uh, G1, B1, dn);
      load_row<NCH>(X + t * d, s, d, xr);
      row_stats<NCH>(xr, s, d, m, rx); normalize<NCH>(xr, s, d, m, rx, xh); affine<NCH>(xh, G2, B2, sn);
      Row ddn, dsn, tmp, du, dhh, dh, dxs;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float a = dn.v[c].x - sn.v[c].x, e = dn.v[c].y - sn.v[c].y, f = dn.v[c].z - sn.v[c].z, g = dn.v[c].w - sn.v[c].w;
        part[6].v[c].x += a * a * dout; part[6].v[c].y += e * e * dout; part[6].v[c].z += f * f * dout; part[6].v[c].w += g * g * dout;
        ddn.v[c].x = 2.f * a * Wc.v[c].x * dout; ddn.v[c].y = 2.f * e * Wc.v[c].y * dout;
        ddn.v[c].z = 2.f * f * Wc.v[c].z * dout; ddn.v[c].w = 2.f * g * Wc.v[c].w * dout;
        dsn.v[c].x = -ddn.v[c].x; dsn.v[c].y = -ddn.v[c].y; dsn.v[c].z = -ddn.v[c].z; dsn.v[c].w = -ddn.v[c].w;
      }
      if (s == 0) part_bc += dout;
      // layer_norm1 (dynamic): dn = uh*g1 + b1
      acc_mul_row<NCH>(part[2], ddn, uh); acc_row<NCH>(part[3], ddn);
      mul_rows<NCH>(ddn, G1, tmp);
      ln_bwd_row<NCH>(tmp, uh, ru, s, d, du);
      // pff_n1.layer_norm: u = hh*gp + bp
      acc_mul_row<NCH>(part[0], du, hh); acc_row<NCH>(part[1], du);
      mul_rows<NCH>(du, Gp, dhh);
      ln_bwd_row<NCH>(dhh, hh, rh, s, d, dh);
      store_row<NCH>(dH2 + t * d, s, d, dh);
      // layer_norm2 (static): sn = xh*g2 + b2
      acc_mul_row<NCH>(part[4], dsn, xh); acc_row<NCH>(part[5], dsn);
      mul_rows<NCH>(dsn, G2, tmp);
      ln_bwd_row<NCH>(tmp, xh, rx, s, d, dxs);
      store_row<NCH>(dXs + t * d, s, d, dxs);
    }
  }
  if (blockIdx.x == 0 && slot == 0) {        // the shared padding token is masked out of every mean: zero gradient rows
    Row zr;
    zero_row<NCH>(zr);
    const int64_t tp = row_off[B];
    store_row<NCH>(dH2 + tp * d, s, d, zr);
    store_row<NCH>(dXs + tp * d, s, d, zr);
  }
  block_colsum_store<NCH, 7>(part, d, slab + (int64_t)blockIdx.x * (7 * d + 1), lds);
  if (s == 0) lds_bc[slot] = part_bc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int t = 0; t < 16; ++t) a += lds_bc[t];
    slab[(int64_t)blockIdx.x * (7 * d + 1) + 7 * d] = a;
  }
}

// sum of row losses / B in a fixed order (one block): losses[0] = bce
// Blocks behind the first: a buffer to zero in the same launch (small batches: the d x_hat rows the backward kernel's heads add into -- the
// one-pass tail reduction used to zero them in front of the backward kernel; it now rides in the launch BEHIND that kernel, fused_bwd.hip).
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ row_loss, int64_t B, float* __restrict__ out, int zero_recon,
                                                           float4* __restrict__ zero_buf, int64_t zero_n4) {
  __shared__ float red[16];
  if (blockIdx.x > 0) {
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < zero_n4; i += (int64_t)(gridDim.x - 1) * 1024) zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  loss_reduce_role<1024>(row_loss, B, out, zero_recon, red);
}

// dst[v][j] += sum_blk slab[blk][v][j]   (v < nv, j < d), blocks ascending; dst pointers per v
struct ColsumDst {
  float* p[8];
};
// Fixed-order parallel reduction (64 outputs x 16 block-lanes per workgroup, like slab_reduce_kernel); an optional
// scalar (the classifier bias gradient) lives behind the nv*d column sums of every block: stride = nv*d + has_scalar.
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ slab, int nblk, int nv, int d, int stride,
                                                             ColsumDst dst) {
  __shared__ float part[16][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s0 = 0.f, s1 = 0.f;
  if (i < stride) {
    int b = q;
    for (; b + 16 < nblk; b += 32) { s0 += slab[(int64_t)b * stride + i]; s1 += slab[(int64_t)(b + 16) * stride + i]; }
    if (b < nblk) s0 += slab[(int64_t)b * stride + i];
  }
  part[q][o] = s0 + s1;
  __syncthreads();
  if (q == 0 && i < stride) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += part[t][o];
    const int v = i / d, j = i - v * d;           // i == nv*d  ->  v == nv, j == 0: the scalar slot
    if (dst.p[v]) dst.p[v][j] += s;
  }
}

// ---- host launchers --------------------------------------------------------------------------------
static inline int nch_of(int d) { return d <= 64 ? 1 : (d <= 128 ? 2 : 4); }

#define DISPATCH_NCH(d, CALL)                      \
  switch (nch_of(d)) {                             \
    case 1: { constexpr int NCH = 1; CALL; } break; \
    case 2: { constexpr int NCH = 2; CALL; } break; \
    default: { constexpr int NCH = 4; CALL; } break; \
  }

int launch_embed_fwd(const int64_t* x, int64_t T, int d, const float* table, const float* dense, const matcha_frozen& f,
                     int n_attr, const float* Wa, const float* ba, float* x0, hipStream_t st, const int32_t* t_dev) {
  if (T <= 0) return MATCHA_OK;
  MATCHA_TRY(check_attr(f, n_attr));
  const AttrSrc attr = attr_src(f, n_attr);
  const int tok_per_blk = T >= 256 * 1024 ? 256 : (T >= 64 * 1024 ? 128 : (T >= 16 * 1024 ? 64 : 16));
  dim3 grid((unsigned)cdiv(T, tok_per_blk));
  const size_t lds = (size_t)n_attr * d * sizeof(float);
  // algorithmic bytes per token: index 8 + embedding row 4d + attribute row 4*n_attr read (attr_mode 1: nothing), x0 row 4d written
  ProfScope ps(MATCHA_PROF_EMBED_FWD, (double)T * (8.0 + 4.0 * d + (attr.mode == 1 ? 0.0 : 4.0 * n_attr) + 4.0 * d), st);
  DISPATCH_NCH(d, hipLaunchKernelGGL((embed_fwd_kernel<NCH>), grid, dim3(256), lds, st, x, T, d, table, dense, attr, n_attr, Wa, ba, x0, tok_per_blk, t_dev));
  MATCHA_CHECK_LAUNCH("embed_fwd_kernel");
  return MATCHA_OK;
}

int launch_gather_rows(const int64_t* ids, int64_t T, int d, const float* table, int64_t n_nodes, float* rows, int32_t* status, hipStream_t st) {
  if (T <= 0) return MATCHA_OK;
  dim3 grid((unsigned)cdiv(T, 32));
  // ALGORITHMIC gather read bytes = 4d + 8 per row (the rows are also written, 4d more, because this entry
  // point materialises them; bench.py reports the read fraction and the read + write total separately)
  ProfScope ps(MATCHA_PROF_GATHER_ROWS, (double)T * (8.0 + 4.0 * d), st);
  DISPATCH_NCH(d, hipLaunchKernelGGL((gather_rows_kernel<NCH>), grid, dim3(256), 0, st, ids, T, d, table, n_nodes, rows, status));
  MATCHA_CHECK_LAUNCH("gather_rows_kernel");
  return MATCHA_OK;
}

int launch_check_ids(const int64_t* ids, int64_t T, int64_t n_nodes, int32_t* status, hipStream_t st) {
  if (T <= 0 || !status) return MATCHA_OK;
  hipLaunchKernelGGL(check_ids_kernel, dim3((unsigned)cdiv(T, 256)), dim3(256), 0, st, ids, T, n_nodes, status);
  MATCHA_CHECK_LAUNCH("check_ids_kernel");
  return MATCHA_OK;
}

int launch_expand_embedding(const int64_t* x, int64_t B, int L, int d, const int32_t* row_off, const float* H2, const float* X, const float* gp,
                            const float* bp, float* dynamic, float* static_, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  dim3 grid((unsigned)cdiv(B * L, 16));
  DISPATCH_NCH(d, hipLaunchKernelGGL((expand_embedding_kernel<NCH>), grid, dim3(256), 0, st, x, B, L, d, row_off, H2, X, gp, bp, dynamic, static_));
  MATCHA_CHECK_LAUNCH("expand_embedding_kernel");
  return MATCHA_OK;
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int64_t n) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n && (reinterpret_cast<uintptr_t>(p + i4) & 15) == 0) {
    *reinterpret_cast<float4*>(p + i4) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int64_t i = i4; i < n && i < i4 + 4; ++i) p[i] = 0.f;
  }
}
int zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return MATCHA_OK;
  MATCHA_CHECK_ARG(p && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, "zero_async: unaligned buffer");
  const int64_t n = (int64_t)(bytes / 4);
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)cdiv(cdiv(n, 4), 256)), dim3(256), 0, st, (float*)p, n);
  MATCHA_CHECK_LAUNCH("zero_f32_kernel");
  return MATCHA_OK;
}

int launch_fill_i32(int32_t* p, int n, int32_t v, hipStream_t st) {
  if (n <= 0) return MATCHA_OK;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, p, n, v);
  MATCHA_CHECK_LAUNCH("fill_i32_kernel");
  return MATCHA_OK;
}

int launch_embed_scatter(const int64_t* x, int64_t T, int d, const float* dx0, float* dtable, hipStream_t st, const int32_t* t_dev) {
  if (T <= 0) return MATCHA_OK;
  ProfScope ps(MATCHA_PROF_EMBED_SCATTER, (double)T * (8.0 + 8.0 * d), st);   // read dx0 row + index, add 4d bytes
  hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)cdiv(T * d, 256)), dim3(256), 0, st, x, T, d, dx0, dtable, t_dev);
  MATCHA_CHECK_LAUNCH("embed_scatter_kernel");
  return MATCHA_OK;
}

int launch_ln3_fwd(const float* X, int64_t T, int d, const float* gq, const float* bq, const float* gk, const float* bk,
                   const float* gv, const float* bv, float* qin, float* kin, float* vin, float* stats, hipStream_t st, const int32_t* t_dev) {
  if (T <= 0) return MATCHA_OK;
  dim3 grid((unsigned)cdiv(T, 16));
  ProfScope ps(MATCHA_PROF_LN3_FWD, (double)T * 16.0 * d, st);      // read X, write qin/kin/vin
  DISPATCH_NCH(d, hipLaunchKernelGGL((ln3_fwd_kernel<NCH>), grid, dim3(256), 0, st, X, T, d, gq, bq, gk, bk, gv, bv, qin, kin, vin, stats, t_dev));
  MATCHA_CHECK_LAUNCH("ln3_fwd_kernel");
  return MATCHA_OK;
}

// number of blocks used by the column-sum kernels for n items (tokens / rows)
int colsum_blocks(int64_t n, int* per_blk) {
  int64_t blocks = cdiv(n, 64);
  if (blocks > 512) blocks = 512;                   // 2 workgroups per CU; fewer slabs to reduce
  if (blocks < 1) blocks = 1;
  int64_t per = cdiv(cdiv(n, blocks), 16) * 16;
  if (per < 16) per = 16;
  *per_blk = (int)per;
  return (int)cdiv(n, per);
}

int launch_ln3_bwd(const float* X, const float* dqin, const float* dkin, const float* dvin, const float* dXs, int64_t T, int d,
                   const float* gq, const float* gk, const float* gv, float* dZ0, float* slab, float* dgq, float* dbq,
                   float* dgk, float* dbk, float* dgv, float* dbv, hipStream_t st, const int32_t* t_dev) {
  if (T <= 0) return MATCHA_OK;
  int per;
  const int nblk = colsum_blocks(T, &per);
  ProfScope ps(MATCHA_PROF_LN3_BWD, (double)T * 24.0 * d, st);      // read X, dqin, dkin, dvin, dXs; write dZ0
  DISPATCH_NCH(d, hipLaunchKernelGGL((ln3_bwd_kernel<NCH>), dim3(nblk), dim3(256), 0, st, X, dqin, dkin, dvin, dXs, T, d, gq, gk, gv, dZ0, slab, per, t_dev));
  MATCHA_CHECK_LAUNCH("ln3_bwd_kernel");
  ColsumDst dst = {{dgq, dbq, dgk, dbk, dgv, dbv, nullptr, nullptr}};
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)cdiv(6 * d, 64)), dim3(1024), 0, st, slab, nblk, 6, d, 6 * d, dst);
  MATCHA_CHECK_LAUNCH("colsum_reduce_kernel");
  return MATCHA_OK;
}

int launch_loss_reduce(const float* row_loss, int64_t B, float* bce_out, hipStream_t st, bool zero_recon, float* zero_buf, size_t zero_bytes) {
  MATCHA_CHECK_ARG(zero_bytes % 16 == 0 && (uintptr_t)zero_buf % 16 == 0, "loss_reduce: zero_buf");
  const int64_t n4 = zero_buf ? (int64_t)(zero_bytes / 16) : 0;
  int zb = (int)cdiv(n4, 4096);                           // four float4 per thread
  if (zb > 64) zb = 64;
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1 + zb), dim3(1024), 0, st, row_loss, B, bce_out, zero_recon ? 1 : 0, reinterpret_cast<float4*>(zero_buf), n4);
  MATCHA_CHECK_LAUNCH("loss_reduce_kernel");
  return MATCHA_OK;
}

int launch_head_fwd(const int32_t* row_off, const float* H2, const float* X, int64_t B, int L, int d, const HeadParams& hp,
                    const float* y, const float* w, float* logits, float* row_loss, float* bce_out, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  float* rl = (y && w) ? row_loss : nullptr;
  ProfScope ps(MATCHA_PROF_HEAD_FWD, (double)B * L * (8.0 * d + 8.0), st);   // read H2, X rows + ids
  DISPATCH_NCH(d, hipLaunchKernelGGL((head_fwd_kernel<NCH>), dim3((unsigned)cdiv(B, 16)), dim3(256), 0, st, row_off, H2, X, B, L, d, hp, y, w, logits, rl));
  MATCHA_CHECK_LAUNCH("head_fwd_kernel");
  if (rl && bce_out) {
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, st, rl, B, bce_out, 0, (float4*)nullptr, (int64_t)0);
    MATCHA_CHECK_LAUNCH("loss_reduce_kernel");
  }
  return MATCHA_OK;
}

int launch_head_bwd(const int32_t* row_off, const float* H2, const float* X, int64_t B, int L, int d, const HeadParams& hp,
                    const float* y, const float* w, const float* logits, const float* dlogits, float alpha, float* dH2,
                    float* dXs, float* slab, const HeadParams& ghp, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  int per;
  const int nblk = colsum_blocks(B, &per);
  {
    ProfScope ps(MATCHA_PROF_HEAD_BWD, (double)B * L * (16.0 * d + 8.0), st);  // read H2, X; write dH2, dXs
    DISPATCH_NCH(d, hipLaunchKernelGGL((head_bwd_kernel<NCH>), dim3(nblk), dim3(256), 0, st, row_off, H2, X, B, L, d, hp, y, w, logits, dlogits, alpha, dH2, dXs, slab, per));
  }
  MATCHA_CHECK_LAUNCH("head_bwd_kernel");
  ColsumDst dst = {{(float*)ghp.gp, (float*)ghp.bp, (float*)ghp.g1, (float*)ghp.b1, (float*)ghp.g2, (float*)ghp.b2, (float*)ghp.wc, (float*)ghp.bc}};
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)cdiv(7 * d + 1, 64)), dim3(1024), 0, st, slab, nblk, 7, d, 7 * d + 1, dst);
  MATCHA_CHECK_LAUNCH("colsum_reduce_kernel");
  return MATCHA_OK;
}

size_t colsum_slab_bytes(int64_t n, int nv, int d) {
  int per;
  const int nblk = colsum_blocks(n, &per);
  return align_up(((size_t)nblk * nv * d + nblk) * sizeof(float), 256);
}

}  // namespace matcha

using namespace matcha;

extern "C" int matcha_embed_fwd(const int64_t* x, int64_t T, int32_t d, const float* table, const float* dense,
                                const float* attr_table, int32_t n_attr, const float* attr_w, const float* attr_b,
                                float* x0, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(x && attr_table && attr_w && attr_b && x0, "matcha_embed_fwd: null pointer");
  MATCHA_CHECK_ARG(d % 4 == 0 && d > 0 && d <= 256, "matcha_embed_fwd: d=%d must be a multiple of 4, <= 256", d);
  matcha_frozen f;
  memset(&f, 0, sizeof(f));
  f.attr_table = attr_table;                      // op-level entry point: plain [N+1, n_attr] rows
  return launch_embed_fwd(x, T, d, table, dense, f, n_attr, attr_w, attr_b, x0, (hipStream_t)stream, nullptr);
}

extern "C" int matcha_embed_scatter_bwd(const int64_t* x, int64_t T, int32_t d, const float* dx0, float* dtable,
                                        matcha_stream_t stream) {
  MATCHA_CHECK_ARG(x && dx0 && dtable, "matcha_embed_scatter_bwd: null pointer");
  return launch_embed_scatter(x, T, d, dx0, dtable, (hipStream_t)stream, nullptr);
}

extern "C" int matcha_ln3_fwd(const float* X, int64_t T, int32_t d, const float* gq, const float* bq, const float* gk,
                              const float* bk, const float* gv, const float* bv, float* qin, float* kin, float* vin,
                              float* stats, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(X && gq && bq && gk && bk && gv && bv && qin && kin && vin, "matcha_ln3_fwd: null pointer");
  MATCHA_CHECK_ARG(d % 4 == 0 && d > 0 && d <= 256, "matcha_ln3_fwd: d=%d must be a multiple of 4, <= 256", d);
  return launch_ln3_fwd(X, T, d, gq, bq, gk, bk, gv, bv, qin, kin, vin, stats, (hipStream_t)stream, nullptr);
}
