// Token-level (HBM-bound) kernels of the hyperedge classifier: embedding-row gather + attribute path,
// the shared-statistics LayerNorm in front of Q/K/V, the classifier tail with the per-hyperedge segmented
// reduction + weighted BCE, and their backward passes.
//
// Layout: activations are [T = B*L tokens, d] fp32 row-major; a token row is handled by a group of 16
// adjacent lanes, each owning float4 chunks j = 4*s + 64*c (s = lane & 15), so one wave-instruction reads
// four whole rows (4 x 256 B at d = 64) and every row statistic is a 4-step xor-shuffle reduction.
#include "kernels.hpp"

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
                                                        const float* __restrict__ attr_table, int n_attr,
                                                        const float* __restrict__ Wa, const float* __restrict__ ba,
                                                        float* __restrict__ x0) {
  const int s = threadIdx.x & 15;
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (t >= T) return;
  const int64_t id = x[t];
  Row e;
  if (table) load_row<NCH>(table + id * d, s, d, e);
  else if (dense) load_row<NCH>(dense + t * d, s, d, e);
  else zero_row<NCH>(e);
  Row out;
  load_row<NCH>(ba, s, d, out);
  const float* arow = attr_table + id * n_attr;
  for (int c0 = 0; c0 < n_attr; ++c0) {
    const float a = arow[c0];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j = 4 * s + 64 * c;
      if (j < d) {
        out.v[c].x += a * Wa[(j + 0) * n_attr + c0];
        out.v[c].y += a * Wa[(j + 1) * n_attr + c0];
        out.v[c].z += a * Wa[(j + 2) * n_attr + c0];
        out.v[c].w += a * Wa[(j + 3) * n_attr + c0];
      }
    }
  }
  acc_row<NCH>(out, e);
  store_row<NCH>(x0 + t * d, s, d, out);
}

// K1 alone: rows[t] = table[ids[t]]   (Wrap_Embedding.forward, Modules.py:33-34; save_embeddings main.py:471)
template <int NCH>
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, int64_t T, int d,
                                                          const float* __restrict__ table, float* __restrict__ rows) {
  const int s = threadIdx.x & 15;
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (t >= T) return;
  Row e;
  load_row<NCH>(table + ids[t] * d, s, d, e);
  store_row<NCH>(rows + t * d, s, d, e);
}

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// dtable[x[t]] += dx0[t]  (row 0 = padding_idx never receives a gradient)
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ x, int64_t T, int d,
                                                            const float* __restrict__ dx0, float* __restrict__ dtable) {
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
                                                      float* __restrict__ kin, float* __restrict__ vin, float* __restrict__ stats) {
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
                                                      float* __restrict__ dZ0, float* __restrict__ slab, int tok_per_blk) {
  __shared__ float lds[16 * 256];
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
__global__ __launch_bounds__(256) void head_fwd_kernel(const int64_t* __restrict__ x, const float* __restrict__ H2,
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
  for (int l = 0; l < L; ++l) {
    const int64_t t = b * L + l;
    if (x[t] == 0) continue;                          // uniform within the 16-lane group
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
__global__ __launch_bounds__(256) void head_bwd_kernel(const int64_t* __restrict__ x, const float* __restrict__ H2,
                                                       const float* __restrict__ X, int64_t B, int L, int d, HeadParams hp,
                                                       const float* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ logits, const float* __restrict__ dlogits,
                                                       float alpha, float* __restrict__ dH2, float* __restrict__ dXs,
                                                       float* __restrict__ slab, float* __restrict__ slab_bc, int rows_per_blk) {
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
    float cnt = 0.f;
    for (int l = 0; l < L; ++l) cnt += (x[b * L + l] != 0) ? 1.f : 0.f;
    float dz;
    if (dlogits) dz = dlogits[b];
    else {
      const float z = logits[b];
      const float sg = 1.f / (1.f + expf(-z));
      dz = alpha * w[b] * (sg - y[b]) / (float)B;
    }
    const float dout = dz / (cnt + 1e-15f);
    for (int l = 0; l < L; ++l) {
      const int64_t t = b * L + l;
      Row zr;
      zero_row<NCH>(zr);
      if (x[t] == 0) {                               // pads: masked out of the mean -> zero gradient
        store_row<NCH>(dH2 + t * d, s, d, zr);
        store_row<NCH>(dXs + t * d, s, d, zr);
        continue;
      }
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
  block_colsum_store<NCH, 7>(part, d, slab + (int64_t)blockIdx.x * 7 * d, lds);
  if (s == 0) lds_bc[slot] = part_bc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int t = 0; t < 16; ++t) a += lds_bc[t];
    slab_bc[blockIdx.x] = a;
  }
}

// sum of row losses / B in a fixed order (one block): losses[0] = bce
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ row_loss, int64_t B, float* __restrict__ out) {
  __shared__ float red[256];
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) a += row_loss[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)B;
}

// dst[v][j] += sum_blk slab[blk][v][j]   (v < nv, j < d), blocks ascending; dst pointers per v
struct ColsumDst {
  float* p[8];
};
__global__ void colsum_reduce_kernel(const float* __restrict__ slab, int nblk, int nv, int d, ColsumDst dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv * d) return;
  float a = 0.f;
  for (int b = 0; b < nblk; ++b) a += slab[(int64_t)b * nv * d + i];
  const int v = i / d, j = i - v * d;
  if (dst.p[v]) dst.p[v][j] += a;
}
__global__ void scalar_reduce_kernel(const float* __restrict__ slab, int n, float* dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float a = 0.f;
    for (int i = 0; i < n; ++i) a += slab[i];
    dst[0] += a;
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

int launch_embed_fwd(const int64_t* x, int64_t T, int d, const float* table, const float* dense, const float* attr_table,
                     int n_attr, const float* Wa, const float* ba, float* x0, hipStream_t st) {
  if (T <= 0) return MATCHA_OK;
  dim3 grid((unsigned)cdiv(T, 16));
  // algorithmic bytes per token: index 8 + embedding row 4d + attribute row 4*n_attr read, x0 row 4d written
  ProfScope ps(MATCHA_PROF_EMBED_FWD, (double)T * (8.0 + 4.0 * d + 4.0 * n_attr + 4.0 * d), st);
  DISPATCH_NCH(d, hipLaunchKernelGGL((embed_fwd_kernel<NCH>), grid, dim3(256), 0, st, x, T, d, table, dense, attr_table, n_attr, Wa, ba, x0));
  MATCHA_CHECK_LAUNCH("embed_fwd_kernel");
  return MATCHA_OK;
}

int launch_gather_rows(const int64_t* ids, int64_t T, int d, const float* table, float* rows, hipStream_t st) {
  if (T <= 0) return MATCHA_OK;
  dim3 grid((unsigned)cdiv(T, 16));
  // gather read bytes = 4d + 8 per row (+ 4d written because the rows are materialised here)
  ProfScope ps(MATCHA_PROF_GATHER_ROWS, (double)T * (8.0 + 8.0 * d), st);
  DISPATCH_NCH(d, hipLaunchKernelGGL((gather_rows_kernel<NCH>), grid, dim3(256), 0, st, ids, T, d, table, rows));
  MATCHA_CHECK_LAUNCH("gather_rows_kernel");
  return MATCHA_OK;
}

int launch_fill_i32(int32_t* p, int n, int32_t v, hipStream_t st) {
  if (n <= 0) return MATCHA_OK;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, p, n, v);
  MATCHA_CHECK_LAUNCH("fill_i32_kernel");
  return MATCHA_OK;
}

int launch_embed_scatter(const int64_t* x, int64_t T, int d, const float* dx0, float* dtable, hipStream_t st) {
  if (T <= 0) return MATCHA_OK;
  ProfScope ps(MATCHA_PROF_EMBED_SCATTER, (double)T * (8.0 + 8.0 * d), st);   // read dx0 row + index, add 4d bytes
  hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)cdiv(T * d, 256)), dim3(256), 0, st, x, T, d, dx0, dtable);
  MATCHA_CHECK_LAUNCH("embed_scatter_kernel");
  return MATCHA_OK;
}

int launch_ln3_fwd(const float* X, int64_t T, int d, const float* gq, const float* bq, const float* gk, const float* bk,
                   const float* gv, const float* bv, float* qin, float* kin, float* vin, float* stats, hipStream_t st) {
  if (T <= 0) return MATCHA_OK;
  dim3 grid((unsigned)cdiv(T, 16));
  ProfScope ps(MATCHA_PROF_LN3_FWD, (double)T * 16.0 * d, st);      // read X, write qin/kin/vin
  DISPATCH_NCH(d, hipLaunchKernelGGL((ln3_fwd_kernel<NCH>), grid, dim3(256), 0, st, X, T, d, gq, bq, gk, bk, gv, bv, qin, kin, vin, stats));
  MATCHA_CHECK_LAUNCH("ln3_fwd_kernel");
  return MATCHA_OK;
}

// number of blocks used by the column-sum kernels for n items (tokens / rows)
int colsum_blocks(int64_t n, int* per_blk) {
  int64_t blocks = cdiv(n, 64);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  int64_t per = cdiv(cdiv(n, blocks), 16) * 16;
  if (per < 16) per = 16;
  *per_blk = (int)per;
  return (int)cdiv(n, per);
}

int launch_ln3_bwd(const float* X, const float* dqin, const float* dkin, const float* dvin, const float* dXs, int64_t T, int d,
                   const float* gq, const float* gk, const float* gv, float* dZ0, float* slab, float* dgq, float* dbq,
                   float* dgk, float* dbk, float* dgv, float* dbv, hipStream_t st) {
  if (T <= 0) return MATCHA_OK;
  int per;
  const int nblk = colsum_blocks(T, &per);
  ProfScope ps(MATCHA_PROF_LN3_BWD, (double)T * 24.0 * d, st);      // read X, dqin, dkin, dvin, dXs; write dZ0
  DISPATCH_NCH(d, hipLaunchKernelGGL((ln3_bwd_kernel<NCH>), dim3(nblk), dim3(256), 0, st, X, dqin, dkin, dvin, dXs, T, d, gq, gk, gv, dZ0, slab, per));
  MATCHA_CHECK_LAUNCH("ln3_bwd_kernel");
  ColsumDst dst = {{dgq, dbq, dgk, dbk, dgv, dbv, nullptr, nullptr}};
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)cdiv(6 * d, 256)), dim3(256), 0, st, slab, nblk, 6, d, dst);
  MATCHA_CHECK_LAUNCH("colsum_reduce_kernel");
  return MATCHA_OK;
}

int launch_head_fwd(const int64_t* x, const float* H2, const float* X, int64_t B, int L, int d, const HeadParams& hp,
                    const float* y, const float* w, float* logits, float* row_loss, float* bce_out, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  float* rl = (y && w) ? row_loss : nullptr;
  ProfScope ps(MATCHA_PROF_HEAD_FWD, (double)B * L * (8.0 * d + 8.0), st);   // read H2, X rows + ids
  DISPATCH_NCH(d, hipLaunchKernelGGL((head_fwd_kernel<NCH>), dim3((unsigned)cdiv(B, 16)), dim3(256), 0, st, x, H2, X, B, L, d, hp, y, w, logits, rl));
  MATCHA_CHECK_LAUNCH("head_fwd_kernel");
  if (rl && bce_out) {
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, rl, B, bce_out);
    MATCHA_CHECK_LAUNCH("loss_reduce_kernel");
  }
  return MATCHA_OK;
}

int launch_head_bwd(const int64_t* x, const float* H2, const float* X, int64_t B, int L, int d, const HeadParams& hp,
                    const float* y, const float* w, const float* logits, const float* dlogits, float alpha, float* dH2,
                    float* dXs, float* slab, const HeadParams& ghp, hipStream_t st) {
  if (B <= 0) return MATCHA_OK;
  int per;
  const int nblk = colsum_blocks(B, &per);
  float* slab_bc = slab + (size_t)nblk * 7 * d;
  ProfScope ps(MATCHA_PROF_HEAD_BWD, (double)B * L * (16.0 * d + 8.0), st);  // read H2, X; write dH2, dXs
  DISPATCH_NCH(d, hipLaunchKernelGGL((head_bwd_kernel<NCH>), dim3(nblk), dim3(256), 0, st, x, H2, X, B, L, d, hp, y, w, logits, dlogits, alpha, dH2, dXs, slab, slab_bc, per));
  MATCHA_CHECK_LAUNCH("head_bwd_kernel");
  ColsumDst dst = {{(float*)ghp.gp, (float*)ghp.bp, (float*)ghp.g1, (float*)ghp.b1, (float*)ghp.g2, (float*)ghp.b2, (float*)ghp.wc, nullptr}};
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)cdiv(7 * d, 256)), dim3(256), 0, st, slab, nblk, 7, d, dst);
  MATCHA_CHECK_LAUNCH("colsum_reduce_kernel");
  hipLaunchKernelGGL(scalar_reduce_kernel, dim3(1), dim3(64), 0, st, slab_bc, nblk, (float*)ghp.bc);
  MATCHA_CHECK_LAUNCH("scalar_reduce_kernel");
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
  return launch_embed_fwd(x, T, d, table, dense, attr_table, n_attr, attr_w, attr_b, x0, (hipStream_t)stream);
}

extern "C" int matcha_embed_scatter_bwd(const int64_t* x, int64_t T, int32_t d, const float* dx0, float* dtable,
                                        matcha_stream_t stream) {
  MATCHA_CHECK_ARG(x && dx0 && dtable, "matcha_embed_scatter_bwd: null pointer");
  return launch_embed_scatter(x, T, d, dx0, dtable, (hipStream_t)stream);
}

extern "C" int matcha_ln3_fwd(const float* X, int64_t T, int32_t d, const float* gq, const float* bq, const float* gk,
                              const float* bk, const float* gv, const float* bv, float* qin, float* kin, float* vin,
                              float* stats, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(X && gq && bq && gk && bk && gv && bv && qin && kin && vin, "matcha_ln3_fwd: null pointer");
  MATCHA_CHECK_ARG(d % 4 == 0 && d > 0 && d <= 256, "matcha_ln3_fwd: d=%d must be a multiple of 4, <= 256", d);
  return launch_ln3_fwd(X, T, d, gq, bq, gk, bk, gv, bv, qin, kin, vin, stats, (hipStream_t)stream);
}
