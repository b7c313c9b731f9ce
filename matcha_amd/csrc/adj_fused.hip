// adj-mode front end at embed_dim 64, fused (round 4): MultipleEmbedding.forward (Modules.py:176-201) + the attribute path and next_w
// (Modules.py:263-270) as THREE kernels over the chromosome-sorted token rows of adj_frontend.hip's counting sort:
//
//   adj_fused_fwd_kernel    per 64 sorted rows of one chromosome c: feature rows gathered by node id (SparseEmbedding :67) -> dropout(0.2)
//                           (:186) -> W0_c (d x n_c) -> tanh -> W1_c (d x d) (TiedAutoEncoder :104-113) = the node row (:188), then
//                           + attribute_nn(attribute row) (:263-269) -> next_w -> tanh (:270) = X.  One chain of four MFMA products per
//                           tile; Hs = tanh(.) of the first, TH = tanh(node), x0 and X are the only rows that reach HBM (1 KB per
//                           token; the layer-by-layer path moved 2.3 KB in six launches).
//   adj_recon_kernel        the reconstruction branch (:192-199) for the tokens outside chromosome r: rec = TH Wr^T + br in 64-column
//                           chunks, D = rec - inter[x - 1, columns of r], loss += D^2 -- and, in a training forward, the branch's whole
//                           BACKWARD in the same pass: dTH += D Wr, dWr += D^T TH, dbr += colsum(D).  The [m, n_r] residual
//                           (0.18 GB per 65 536-row step, read and written five times by the layer-by-layer path) never exists; what
//                           leaves the kernel is one d-wide row per token (dnr = dTH (1 - TH^2), UNSCALED: the upstream factor beta /
//                           *drecon enters in the backward kernels) and one [n_r, d + 1] gradient of the head.
//   adj_fused_bwd_kernel    per (chromosome, window of sorted rows): dnode = dX0 (front_bwd_kernel) + g dnr;  dW1_c += dnode^T Hs;
//                           dZ = (dnode W1_c) (1 - Hs^2);  dW0_c += dZ^T (feature rows x dropout mask) -- the three products of a
//                           64-row step back to back on LDS tiles, the weight gradients in MFMA accumulators for the whole window and
//                           added with float atomics at its end (as adj_tn_kernel did; the only non-reproducible sums of the adj path).
//
// Feature rows are read with aligned 16-byte loads: the fused path requires matcha_frozen.feat_row_pad = 64 (rows padded with zeros to
// a multiple of 64 floats, so a K chunk never needs a tail mask either).  n_c and n_r are arbitrary: chunks of 64 columns, column groups
// of 256 (= 4 accumulator tiles per wavefront) walked in turn.
#include <string.h>

#include "adj_common.hpp"
#include "attr_src.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kLd = 68;
constexpr int kTile = 64 * kLd;
constexpr int kAttrCols = 32;

// Work item of the forward kernel: block -> (bucket c, first sorted row, rows).  Bucket c owns ceil(len_c / 64) consecutive items;
// every wavefront finds its block's item with one 64-lane scan over the <= 64 buckets (no launch-time knowledge of the segment
// lengths: the grid is sized for the bound T / 64 + C + 1 and the blocks behind the last item leave).
__device__ __forceinline__ bool find_item64(const int32_t* __restrict__ seg, int nb, int item, int& c, int& p0, int& nrows) {
  const int lane = threadIdx.x & 63;
  const int lo = lane < nb ? seg[lane] : 0, hi = lane < nb ? seg[lane + 1] : 0;
  const int steps = (hi - lo + 63) >> 6;
  int incl = steps;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int excl = incl - steps;
  const unsigned long long bal = __ballot(item >= excl && item < incl);
  if (bal == 0) return false;
  c = __ffsll((long long)bal) - 1;
  const int e = __shfl(excl, c, 64), l = __shfl(lo, c, 64), h = __shfl(hi, c, 64);
  p0 = l + 64 * (item - e);
  nrows = h - p0 < 64 ? h - p0 : 64;
  return true;
}

struct AdjFwdArgs {
  const int64_t* ids;                 // [T] node id of each token (0 = padding)
  const int32_t *order, *seg, *bounds;
  const int64_t* feat_off;
  const float* feats;
  int feat_pad;
  const float *w0, *w1;               // adj_w0 (chromosome c at d * bounds[c], row stride n_c), adj_w1 [C][64][64]
  AttrSrc attr;
  const float *Wa, *ba, *Wn, *bn;
  float *Hs, *TH;                     // sorted-row outputs for the backward pass / the reconstruction branch (null: not written)
  float* node;                        // token-order node rows (matcha_node_embeddings; null: not written)
  float *x0, *X;                      // token-order rows of the encoder's input (null: not written)
  int C;
  const uint64_t* seed;
  float p_drop;
  const int32_t* slot_map;            // token index -> original [B, L] slot (dropout counter); null = identity
};

// HEAD: also run the attribute path + next_w (x0, X); without it the kernel stops at the node rows (get_node_embeddings).
// The kernel is a chain of short dependent phases -- row indices -> feature rows -> four products with a barrier between them (22 us for
// one workgroup alone) -- and what hides that latency is the other workgroups of the CU: FOUR of them (two LDS tiles = 35 KB, <= 128
// registers).  For that the tiles are reused as soon as they are dead, the weights of the small products are fetched a phase before
// their use (not held through the gather-GEMM) and the attribute rows never get a tile: under attr_mode 1 a lane builds its MFMA
// fragments from (chromosome column, coordinate), under attr_mode 0 it reads them from the table.
template <bool HEAD>
__global__ __launch_bounds__(256, 4) void adj_fused_fwd_kernel(AdjFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* T0 = lds;                                // GEMM 1 A tile (feature chunk), then Hs, then node -> x0
  float* T1 = lds + kTile;                        // GEMM 1 B tile (W0 chunk), then TH, then X
  __shared__ int64_t rowoff[64];                  // element offset of the gathered feature row
  __shared__ int rowtok[64];                      // token index of the sorted row, -1 = past the item's rows
  __shared__ uint32_t rowh[64];                   // dropout: lowbias32(slot ^ key) of the row
  __shared__ int rowcol[64];                      // attr_mode 1: chromosome column of the row's attribute row (-1: all zeros)
  __shared__ float rowcoord[64];
  __shared__ int rowid[64];
  __shared__ int abounds[64];                     // attr_mode 1: the chromosome bounds (searched per row)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  // independent loads first: the attribute bounds, the seed
  if (HEAD && g.attr.mode == 1 && tid >= 64 && tid < 64 + g.attr.n_attr) abounds[tid - 64] = g.attr.bounds[tid - 64];
  const bool drop_on = g.p_drop > 0.f;
  uint64_t seedv = 0;
  if (drop_on) seedv = *g.seed;
  int c, p0, nrows;
  if (!find_item64(g.seg, g.C + 1, blockIdx.x, c, p0, nrows)) return;
  const bool pad = c >= g.C;                      // the padding bucket: node row = 0 (Modules.py:178)
  const int cc_ = pad ? 0 : c;
  // one round trip: the chromosome's bounds and feature offset, this thread's sorted row; then the row's node id
  const int lo_raw = g.bounds[cc_], hi_raw = g.bounds[cc_ + 1];
  const int64_t foff = g.feat_off[cc_];
  int tok = 0;
  if (tid < 64) tok = g.order[p0 + (tid < nrows ? tid : 0)];
  const int lo = pad ? 0 : lo_raw, n_c = pad ? 0 : hi_raw - lo_raw;
  const int ldf = feat_ld(n_c, g.feat_pad);
  const bool drop = drop_on && !pad;
  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (drop) { key = rng_key(seedv, kStreamDropAdj); thr = dropout_threshold(g.p_drop); keep_scale = 1.f / (1.f - g.p_drop); }
  float4 w1f[8];                                  // W1_c as B fragments of the second product
  if (tid < 64) {
    const int64_t id = g.ids[tok];
    const int slot = g.slot_map ? g.slot_map[tok] : tok;
    rowtok[tid] = tid < nrows ? tok : -1;
    rowid[tid] = (int)id;
    rowoff[tid] = pad ? 0 : foff + (id - lo - 1) * (int64_t)ldf;
    rowh[tid] = lowbias32((uint32_t)slot ^ key);
  }
  __syncthreads();                                // the row tables (and the attribute bounds) are in LDS
  if (HEAD && g.attr.mode == 1 && tid < 64) {
    int col; float coord;
    attr_decode(g.attr, abounds, rowid[tid], col, coord);
    rowcol[tid] = col; rowcoord[tid] = coord;     // read after the gather-GEMM's barriers ...
  }
  if (HEAD && pad) __syncthreads();               // ... which the padding bucket does not run
  const int col = 32 * wc + r;
  f32x16 acc = {0};
  if (!pad) {
    // ---- GEMM 1: Hs = tanh( (feature rows x dropout mask) . W0_c^T ), contraction in chunks of 64 feature columns; the loads of chunk
    // kc + 64 are in flight while the MFMAs of chunk kc run ----
    const float* W0 = g.w0 + (int64_t)64 * lo;    // [64][n_c] row-major
    float4 ra[4];
    f4u rw[4];
#define AFF_GLOAD(KC)                                                                                    \
  do {                                                                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      ra[i__] = *reinterpret_cast<const float4*>(g.feats + rowoff[srow + 16 * i__] + (KC) + sc4);        \
      rw[i__] = row4_load(W0 + (int64_t)(srow + 16 * i__) * n_c, (KC) + sc4, n_c);                       \
    }                                                                                                    \
  } while (0)
    AFF_GLOAD(0);
    for (int kc = 0; kc < n_c; kc += 64) {
      const int col0 = kc + sc4;
      if (kc) __syncthreads();                    // the previous chunk's MFMAs are done with the tiles
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;
        float4 e = ra[i];
        if (drop) {
          const uint32_t hr = rowh[row];
          e.x = lowbias32((uint32_t)(col0 + 0) ^ hr) >= thr ? e.x * keep_scale : 0.f;      // counter = (token slot, column): rng_u32
          e.y = lowbias32((uint32_t)(col0 + 1) ^ hr) >= thr ? e.y * keep_scale : 0.f;
          e.z = lowbias32((uint32_t)(col0 + 2) ^ hr) >= thr ? e.z * keep_scale : 0.f;
          e.w = lowbias32((uint32_t)(col0 + 3) ^ hr) >= thr ? e.w * keep_scale : 0.f;
        }
        *reinterpret_cast<float4*>(&T0[row * kLd + sc4]) = e;
        float wv[4];
        row4_fix(rw[i], col0, n_c, wv);
        *reinterpret_cast<float4*>(&T1[row * kLd + sc4]) = make_float4(wv[0], wv[1], wv[2], wv[3]);
      }
      __syncthreads();
      if (kc + 64 < n_c) AFF_GLOAD(kc + 64);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const float4 a = *reinterpret_cast<const float4*>(&T0[(32 * wr + r) * kLd + 8 * cc + 4 * h]);
        const float4 b = *reinterpret_cast<const float4*>(&T1[(32 * wc + r) * kLd + 8 * cc + 4 * h]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
#undef AFF_GLOAD
    {                                             // in flight during the barrier, the tanh epilogue and the Hs rows' way out
      const float* W1 = g.w1 + (int64_t)c * 4096;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) w1f[cc] = *reinterpret_cast<const float4*>(W1 + (32 * wc + r) * 64 + 8 * cc + 4 * h);
    }
    __syncthreads();                              // every wave is done with the last chunk: T0 takes Hs
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      T0[row * kLd + col] = fast_tanh(acc[reg]);
    }
    __syncthreads();                              // Hs tile complete
    if (g.Hs) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;
        if (row < nrows) *reinterpret_cast<float4*>(g.Hs + (int64_t)(p0 + row) * 64 + sc4) = *reinterpret_cast<const float4*>(&T0[row * kLd + sc4]);
      }
    }
    // ---- GEMM 2: node = Hs . W1_c^T ----
    acc = (f32x16){0};
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const float4 a = *reinterpret_cast<const float4*>(&T0[(32 * wr + r) * kLd + 8 * cc + 4 * h]);
      const float4 b = w1f[cc];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }
  // weights of the last two products (B fragments) and the attribute fragments of this lane's row: in flight during the stores below
  float4 wn[8], wa[kAttrCols / 8], af[kAttrCols / 8];
  float bav = 0.f, bnv = 0.f;
  if (HEAD) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) wn[cc] = *reinterpret_cast<const float4*>(g.Wn + (32 * wc + r) * 64 + 8 * cc + 4 * h);
    const int arow = 32 * wr + r;
#pragma unroll
    for (int cc = 0; cc < kAttrCols / 8; ++cc) {
      const int q = 8 * cc + 4 * h;
      wa[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
      af[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < g.attr.n_attr) {
        wa[cc] = *reinterpret_cast<const float4*>(g.Wa + (32 * wc + r) * g.attr.n_attr + q);
        if (g.attr.mode == 1) {
          const int cl = rowcol[arow]; const float cd = rowcoord[arow];
          af[cc] = make_float4(attr_elem(q, cl, cd, g.attr.n_attr), attr_elem(q + 1, cl, cd, g.attr.n_attr), attr_elem(q + 2, cl, cd, g.attr.n_attr),
                               attr_elem(q + 3, cl, cd, g.attr.n_attr));
        } else {
          af[cc] = *reinterpret_cast<const float4*>(g.attr.table + (int64_t)rowid[arow] * g.attr.ld + q);
        }
      }
    }
    bav = g.ba[32 * wc + r]; bnv = g.bn[32 * wc + r];
  }
  __syncthreads();                                // every wave is done reading Hs from T0 (pad bucket: the tiles were never used)
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const float v = pad ? 0.f : acc[reg];
    T0[row * kLd + col] = v;
    T1[row * kLd + col] = fast_tanh(v);
  }
  __syncthreads();                                // node and TH tiles complete
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 16 * i;
    if (row >= nrows) continue;
    if (g.TH && !pad) *reinterpret_cast<float4*>(g.TH + (int64_t)(p0 + row) * 64 + sc4) = *reinterpret_cast<const float4*>(&T1[row * kLd + sc4]);
    if (!HEAD && g.node) *reinterpret_cast<float4*>(g.node + (int64_t)rowtok[row] * 64 + sc4) = *reinterpret_cast<const float4*>(&T0[row * kLd + sc4]);
  }
  if (!HEAD) return;
  // ---- GEMM 3: x0 = node + attribute row . Wa^T + ba   (K = 32, A fragments from registers) ----
  acc = (f32x16){0};
#pragma unroll
  for (int cc = 0; cc < kAttrCols / 8; ++cc) {
    const float4 a = af[cc];
    const float4 b = wa[cc];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    T0[row * kLd + col] += acc[reg] + bav;        // in place: this lane owns the element
  }
  __syncthreads();                                // x0 complete; the TH rows have left T1
  if (g.x0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      if (row < nrows) *reinterpret_cast<float4*>(g.x0 + (int64_t)rowtok[row] * 64 + sc4) = *reinterpret_cast<const float4*>(&T0[row * kLd + sc4]);
    }
  }
  // ---- GEMM 4: X = tanh(x0 . Wn^T + bn) ----
  acc = (f32x16){0};
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const float4 a = *reinterpret_cast<const float4*>(&T0[(32 * wr + r) * kLd + 8 * cc + 4 * h]);
    const float4 b = wn[cc];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    T1[row * kLd + col] = fast_tanh(acc[reg] + bnv);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 16 * i;
    if (row < nrows) *reinterpret_cast<float4*>(g.X + (int64_t)rowtok[row] * 64 + sc4) = *reinterpret_cast<const float4*>(&T1[row * kLd + sc4]);
  }
}

// ---- reconstruction branch, forward + (GRAD) backward in one pass ---------------------------------------------------------------------
struct AdjReconArgs {
  const int64_t* ids;
  const int32_t *order, *seg, *counts;
  const float* TH;                    // [sorted rows][64]
  const float *Wr, *br;               // recon head of chromosome r: [n_r][64], [n_r]   (r_dev: the packed tensors' bases, chromosome c at 64 bounds[c] / bounds[c])
  const float* inter;                 // [N][N]
  int64_t n_nodes;
  int r, lo_r, n_r;
  const int32_t *r_dev, *bounds;      // opts->random_chrom_dev: the chromosome is read here (graph replay), its range from bounds
  int C;                              // number of chromosomes: a device-side value outside [0, C) means: no reconstruction branch
  float* dnr;                         // GRAD: [sorted rows][64] = (d loss / d node) of the branch, unscaled
  float *gW, *gb;                     // GRAD: unscaled head gradient [n_r][64], [n_r] (zeroed by the caller; float atomics)
  float* slab;                        // per-workgroup partial sums of the squared residuals
};

template <bool GRAD>
__global__ __launch_bounds__(256, 2) void adj_recon_kernel(AdjReconArgs g) {
  if (g.r_dev) {
    const int rv = *g.r_dev;
    if (rv < 0 || rv >= g.C) {                    // like the host-int path's random_chrom = -1 (adj_scan_kernel clamps the same way): no branch, loss 0
      if (threadIdx.x == 0) g.slab[blockIdx.x] = 0.f;
      return;
    }
    g.r = rv; g.lo_r = g.bounds[g.r]; g.n_r = g.bounds[g.r + 1] - g.lo_r;
    g.Wr += (int64_t)64 * g.lo_r; g.br += g.lo_r;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ts = lds;                                // TH tile
  float* Ws = lds + kTile;                        // Wr chunk [64 columns of r][64]
  float* Dt = lds + 2 * kTile;                    // D chunk [64 rows][64 columns]
  __shared__ int64_t rowoff[64];                  // element offset of the target row's first column inside inter's ROW (id - 1) * N
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  const int m = g.counts[0];
  const int r0 = g.seg[g.r], len_r = g.seg[g.r + 1] - r0;
  const float gs = m > 0 ? 200.f / ((float)m * (float)g.n_r) : 0.f;
  const int nitems = (m + 63) >> 6;
  const int ngroups = (g.n_r + 255) >> 8;
  const int col = 32 * wc + r;
  float lsum = 0.f;
  for (int grp = 0; grp < ngroups; ++grp) {
    const int cg0 = grp * 256;
    const int nch = (g.n_r - cg0 + 63) >> 6 < 4 ? (g.n_r - cg0 + 63) >> 6 : 4;
    f32x16 aW[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aW[q] = (f32x16){0};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    // Register prefetch: the TH rows and node ids of the NEXT item (requested at the top of an item, a whole item of latency cover) and the
    // Wr chunk of the next (item, chunk) pair (requested right after the current one was staged).
    float4 pt[4], pw[4];
    int64_t pid = 1;
#define ARC_ITEM(ITEM)                                                                                   \
  do {                                                                                                   \
    const int jb__ = (ITEM) * 64;                                                                        \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int j__ = jb__ + srow + 16 * i__;                                                            \
      const int jc__ = j__ < m ? j__ : m - 1;                                                            \
      pt[i__] = *reinterpret_cast<const float4*>(g.TH + (int64_t)(jc__ < r0 ? jc__ : jc__ + len_r) * 64 + sc4); \
    }                                                                                                    \
    if (tid < 64) {                                                                                      \
      const int j__ = jb__ + tid < m ? jb__ + tid : m - 1;                                               \
      pid = g.ids[g.order[j__ < r0 ? j__ : j__ + len_r]];                                                \
    }                                                                                                    \
  } while (0)
#define ARC_WCHUNK(Q)                                                                                    \
  do {                                                                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int cr__ = cg0 + 64 * (Q) + srow + 16 * i__;                                                 \
      pw[i__] = *reinterpret_cast<const float4*>(g.Wr + (int64_t)(cr__ < g.n_r ? cr__ : g.n_r - 1) * 64 + sc4); \
    }                                                                                                    \
  } while (0)
    if ((int)blockIdx.x < nitems) { ARC_ITEM(blockIdx.x); ARC_WCHUNK(0); }
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
      const int j0 = item * 64;
      __syncthreads();                            // the previous item is done with the tiles and the row table
      if (tid < 64) rowoff[tid] = (pid - 1) * g.n_nodes;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;
        float4 v = pt[i];
        if (j0 + row >= m) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&Ts[row * kLd + sc4]) = v;
      }
      if (item + (int)gridDim.x < nitems) ARC_ITEM(item + (int)gridDim.x);
      f32x16 aT = {0};                            // dTH quadrant: rows 32 wr.., features 32 wc..
      for (int q = 0; q < nch; ++q) {
        const int c0 = cg0 + 64 * q;              // first column of the chunk inside chromosome r
        if (q) __syncthreads();                   // previous chunk's products are done with Ws / Dt
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cr = c0 + srow + 16 * i;      // Wr row = column of r
          *reinterpret_cast<float4*>(&Ws[(srow + 16 * i) * kLd + sc4]) = cr < g.n_r ? pw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        ARC_WCHUNK(q + 1 < nch ? q + 1 : 0);      // the chunk the next trip (or the next item) stages
        // target values in the accumulator layout: lane (r, h) of quadrant (wr, wc) holds rows 32 wr + (reg & 3) + 8 (reg >> 2) + 4 h, column
        // c0 + 32 wc + r -- per register 32 consecutive floats of two rows; issued before the product that hides them
        float tg[16];
        const int ccol = c0 + col;
        const bool cin = ccol < g.n_r;
        const int ccl = cin ? ccol : g.n_r - 1;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          tg[reg] = g.inter[rowoff[row] + g.lo_r + ccl];
        }
        const float bias = g.br[ccl];
        f32x16 acc = {0};
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const float4 a = *reinterpret_cast<const float4*>(&Ts[(32 * wr + r) * kLd + 8 * cc + 4 * h]);
          const float4 b = *reinterpret_cast<const float4*>(&Ws[(32 * wc + r) * kLd + 8 * cc + 4 * h]);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float dv = (cin && j0 + row < m) ? acc[reg] + bias - tg[reg] : 0.f;
          lsum += dv * dv;
          if (GRAD) Dt[row * kLd + col] = dv * gs;
        }
        if (GRAD) {
          __syncthreads();                        // D chunk complete
          // dTH[row][k] += sum_col D[row][col] Wr[col][k]   (contraction over the chunk's 64 columns: A rows of Dt, B column walk of Ws)
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) {
            const float4 a = *reinterpret_cast<const float4*>(&Dt[(32 * wr + r) * kLd + 8 * cc + 4 * h]);
            const float* wp = &Ws[(8 * cc + 4 * h) * kLd + 32 * wc + r];
            aT = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], aT, 0, 0, 0);
            aT = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[kLd], aT, 0, 0, 0);
            aT = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * kLd], aT, 0, 0, 0);
            aT = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * kLd], aT, 0, 0, 0);
          }
          // dWr[col][k] += sum_row D[row][col] TH[row][k]   (contraction over the 64 rows);  dbr[col] += sum_row D[row][col]
          f32x16 a2 = aW[0];
          if (q == 1) a2 = aW[1]; else if (q == 2) a2 = aW[2]; else if (q == 3) a2 = aW[3];
          float csq = 0.f;
#pragma unroll 8
          for (int mm = 0; mm < 32; ++mm) {
            const int t = 2 * mm + h;
            const float gd = Dt[t * kLd + 32 * wr + r];
            csq += gd;
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gd, Ts[t * kLd + 32 * wc + r], a2, 0, 0, 0);
          }
          if (q == 0) { aW[0] = a2; cs[0] += csq; } else if (q == 1) { aW[1] = a2; cs[1] += csq; }
          else if (q == 2) { aW[2] = a2; cs[2] += csq; } else { aW[3] = a2; cs[3] += csq; }
        }
      }
      if (GRAD) {
        // dnr = dTH (1 - TH^2): through the (dead) D tile to whole 256-byte rows
        __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float t = Ts[row * kLd + col];
          Dt[row * kLd + col] = aT[reg] * (1.f - t * t);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = srow + 16 * i;
          const int j = j0 + row;
          if (j >= m) continue;
          const int p = j < r0 ? j : j + len_r;
          float4 v = *reinterpret_cast<const float4*>(&Dt[row * kLd + sc4]);
          float* dst = g.dnr + (int64_t)p * 64 + sc4;
          if (grp) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
    }
#undef ARC_ITEM
#undef ARC_WCHUNK
    if (GRAD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= nch) break;
        const int c0 = cg0 + 64 * q;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int crow = c0 + 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (crow < g.n_r) atomicAdd(g.gW + (int64_t)crow * 64 + col, aW[q][reg]);
        }
        float csum = cs[q] + __shfl_xor(cs[q], 32, 64);
        if (wc == 0 && h == 0 && c0 + 32 * wr + r < g.n_r) atomicAdd(g.gb + c0 + 32 * wr + r, csum);
      }
    }
  }
  // partial sum of the squared residuals of this workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (tid == 0) g.slab[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// recon_loss = 100 * sum / (m * n_r)  (mean over columns, mean over rows, * 100; Modules.py:199), 0 when m == 0 (:195)
__global__ __launch_bounds__(256) void adj_recon_sum_kernel(const float* __restrict__ slab, int nslab, const int32_t* __restrict__ counts, int n_r,
                                                            float* __restrict__ out, const int32_t* __restrict__ r_dev, const int32_t* __restrict__ bounds, int C) {
  __shared__ float red[256];
  bool valid = true;
  if (r_dev) {
    const int r = *r_dev;
    valid = r >= 0 && r < C;
    if (valid) n_r = bounds[r + 1] - bounds[r];
  }
  const int m = valid ? counts[0] : 0;
  float s = 0.f;
  for (int i = threadIdx.x; i < nslab; i += 256) s += slab[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = m > 0 ? 100.f * red[0] / ((float)m * (float)n_r) : 0.f;
    out[1] = (float)m;                 // losses[2]: rows of the mean (data-parallel weighting)
  }
}

// ---- encoder backward ---------------------------------------------------------------------------------------------------------------------
struct AdjBwdArgs {
  const int64_t* ids;
  const int32_t *order, *seg, *bounds;
  const int64_t* feat_off;
  const float* feats;
  int feat_pad;
  const float* w1;
  const float* dX0;                   // [T][64] token order (front_bwd_kernel)
  const float* dnr;                   // recon part, sorted rows (null: none)
  const float* drecon; float beta;    // upstream factor of the recon part
  const float* Hs;                    // [sorted rows][64]
  float *gW0, *gW1;                   // gradient tensors (float atomics)
  int C, r, steps_per_item;
  const uint64_t* seed;
  float p_drop;
  const int32_t* slot_map;
  const int32_t* r_dev;
  // blocks [n_main, n_main + 1 + n_apply) of the same launch: block n_main writes the step's `touched` flags (which per-chromosome tensors
  // received a gradient: Modules.py:182-183, :195), the others add the reconstruction head's gradient (left unscaled in rgrad by
  // adj_recon_kernel) into recon_w / recon_b -- two launches less per step than as kernels of their own
  int n_main;
  int32_t* touched; const int32_t* counts;
  const float* rgW; const float* rgb; int n_r; float* dWr; float* dbr;       // dWr / dbr: chromosome r's slices, or the packed bases with r_dev
};

// work item = (chromosome, window of steps_per_item 64-row steps); the grid is sized for the bound and blocks behind the last item leave
__device__ __forceinline__ bool find_window(const int32_t* __restrict__ seg, int nb, int item, int spi, int& c, int& p_lo, int& p_hi) {
  const int lane = threadIdx.x & 63;
  const int lo = lane < nb ? seg[lane] : 0, hi = lane < nb ? seg[lane + 1] : 0;
  const int wins = (hi - lo + 64 * spi - 1) / (64 * spi);
  int incl = wins;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int excl = incl - wins;
  const unsigned long long bal = __ballot(item >= excl && item < incl);
  if (bal == 0) return false;
  c = __ffsll((long long)bal) - 1;
  const int e = __shfl(excl, c, 64), l = __shfl(lo, c, 64), hh = __shfl(hi, c, 64);
  p_lo = l + 64 * spi * (item - e);
  p_hi = p_lo + 64 * spi < hh ? p_lo + 64 * spi : hh;
  return true;
}

__global__ __launch_bounds__(256, 2) void adj_fused_bwd_kernel(AdjBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ws = lds;                                // W1_c [j][k]
  float* Ds = lds + kTile;                        // dnode tile, then feature chunks (even)
  float* Hs_s = lds + 2 * kTile;                  // Hs tile, then feature chunks (odd)
  float* Zs = lds + 3 * kTile;                    // dZ tile
  __shared__ int64_t rowoff[64];
  __shared__ uint32_t rowh[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  if ((int)blockIdx.x >= g.n_main) {
    const int e = (int)blockIdx.x - g.n_main;
    int rc = g.r;
    if (g.r_dev) { const int rv = *g.r_dev; rc = (rv >= 0 && rv < g.C) ? rv : -1; }
    if (e == 0) {
      if (g.touched && tid < 64) {
        if (tid == 0) { g.touched[0] = 1; g.touched[1] = 0; }
        if (tid < g.C) {
          g.touched[2 + tid] = g.seg[tid + 1] > g.seg[tid] ? 1 : 0;
          g.touched[2 + g.C + tid] = (tid == rc && g.counts[0] > 0) ? 1 : 0;
        }
      }
      return;
    }
    if (!g.rgW || rc < 0) return;
    int n_r = g.n_r;
    float* dW = g.dWr;
    float* db = g.dbr;
    if (g.r_dev) { const int lo_r = g.bounds[rc]; n_r = g.bounds[rc + 1] - lo_r; dW += (int64_t)64 * lo_r; db += lo_r; }
    const float gs = g.drecon ? g.drecon[0] : g.beta;
    const int i = (e - 1) * 256 + tid;
    if (i < n_r * 64) dW[i] += gs * g.rgW[i];
    if (i < n_r) db[i] += gs * g.rgb[i];
    return;
  }
  int c, p_lo, p_hi;
  if (!find_window(g.seg, g.C, blockIdx.x, g.steps_per_item, c, p_lo, p_hi)) return;
  const int lo = g.bounds[c], n_c = g.bounds[c + 1] - lo;
  const int ldf = feat_ld(n_c, g.feat_pad);
  const bool drop = g.p_drop > 0.f;
  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (drop) { key = rng_key(*g.seed, kStreamDropAdj); thr = dropout_threshold(g.p_drop); keep_scale = 1.f / (1.f - g.p_drop); }
  int r_chrom = g.r;
  bool r_valid = true;
  if (g.r_dev) { r_chrom = *g.r_dev; r_valid = r_chrom >= 0 && r_chrom < g.C; }      // outside [0, C): no reconstruction branch ran (adj_recon_kernel)
  const bool with_r = g.dnr != nullptr && r_valid && c != r_chrom;
  const float gsc = with_r ? (g.drecon ? g.drecon[0] : g.beta) : 0.f;
  {
    const float* W1 = g.w1 + (int64_t)c * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(&Ws[(srow + 16 * i) * kLd + sc4]) = *reinterpret_cast<const float4*>(W1 + (srow + 16 * i) * 64 + sc4);
  }
  const int col = 32 * wc + r;
  const int64_t foff_c = g.feat_off[c];
  const int ngroups = (n_c + 255) >> 8;
  for (int grp = 0; grp < ngroups; ++grp) {
    const int cg0 = grp * 256;
    const int nch = (n_c - cg0 + 63) >> 6 < 4 ? (n_c - cg0 + 63) >> 6 : 4;
    f32x16 aW0[4], aW1 = {0};
#pragma unroll
    for (int q = 0; q < 4; ++q) aW0[q] = (f32x16){0};
    // Register prefetch one step ahead: the rows of step p0 + 64 (dX0 by token, Hs and the recon part by sorted row) are requested after
    // the dZ barrier of step p0 and land during its feature-chunk products; the row indices they depend on (sorted row -> token -> node
    // id) are requested one step earlier still.  Indices past the window are clamped to its last row and masked when staged.
    float4 pd[4], ph[4], rf[4];                   // rf: feature-chunk prefetch inside a step; between steps it carries the recon rows
    int ptok[4], ttok = 0;
    int64_t tid_id = 0;
#define AFB_IDX(P0)                                                                                      \
  do {                                                                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int p__ = (P0) + srow + 16 * i__;                                                            \
      ptok[i__] = g.order[p__ < p_hi ? p__ : p_hi - 1];                                                  \
    }                                                                                                    \
    if (tid < 64) {                                                                                      \
      const int p__ = (P0) + tid;                                                                        \
      ttok = g.order[p__ < p_hi ? p__ : p_hi - 1];                                                       \
      tid_id = g.ids[ttok];                                                                              \
    }                                                                                                    \
  } while (0)
#define AFB_ROWS(P0)                                                                                     \
  do {                                                                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int p__ = (P0) + srow + 16 * i__;                                                            \
      const int pc__ = p__ < p_hi ? p__ : p_hi - 1;                                                      \
      pd[i__] = *reinterpret_cast<const float4*>(g.dX0 + (int64_t)ptok[i__] * 64 + sc4);                 \
      ph[i__] = *reinterpret_cast<const float4*>(g.Hs + (int64_t)pc__ * 64 + sc4);                       \
    }                                                                                                    \
  } while (0)
#define AFB_RECON(P0)                                                                                    \
  do {                                                                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int p__ = (P0) + srow + 16 * i__;                                                            \
      rf[i__] = *reinterpret_cast<const float4*>(g.dnr + (int64_t)(p__ < p_hi ? p__ : p_hi - 1) * 64 + sc4); \
    }                                                                                                    \
  } while (0)
    AFB_IDX(p_lo);
    AFB_ROWS(p_lo);
    if (with_r) AFB_RECON(p_lo);
    for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
      const int nrows = p_hi - p0 < 64 ? p_hi - p0 : 64;
      __syncthreads();                            // the previous step is done with every tile and the row tables
      // per-lane indices re-derived from an opaque copy of the thread id: left alone the compiler hoists the loop-invariant LDS / global
      // addresses of the whole step out of the loop and spills them (fused_bwd.hip met the same)
      int tid_ = threadIdx.x;
      asm volatile("" : "+v"(tid_));
      const int tid = tid_, lane = tid_ & 63, wave = tid_ >> 6;
      const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
      const int srow = tid_ >> 4, sc4 = (tid_ & 15) * 4;
      const int col = 32 * wc + r;
      if (tid < 64) {
        rowoff[tid] = foff_c + (tid_id - lo - 1) * (int64_t)ldf;
        rowh[tid] = lowbias32((uint32_t)(g.slot_map ? g.slot_map[ttok] : ttok) ^ key);
      }
      // dnode = dX0[token] + g * dnr[sorted row];  Hs
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;
        float4 d = pd[i], hv = ph[i];
        if (with_r) { d.x += gsc * rf[i].x; d.y += gsc * rf[i].y; d.z += gsc * rf[i].z; d.w += gsc * rf[i].w; }
        if (row >= nrows) { d = make_float4(0.f, 0.f, 0.f, 0.f); hv = d; }
        *reinterpret_cast<float4*>(&Ds[row * kLd + sc4]) = d;
        *reinterpret_cast<float4*>(&Hs_s[row * kLd + sc4]) = hv;
      }
      const bool more = p0 + 64 < p_hi;
      if (more) AFB_IDX(p0 + 64);                 // indices of the next step (its rows are requested below, a product later)
      __syncthreads();
      // feature chunk 0 of this group: in flight during the two small products
#define AFB_GLOAD(Q)                                                                                     \
  do {                                                                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__)                                                  \
      rf[i__] = *reinterpret_cast<const float4*>(g.feats + rowoff[srow + 16 * i__] + cg0 + 64 * (Q) + sc4); \
  } while (0)
      AFB_GLOAD(0);
      // dW1[j][k] += sum_rows dnode[row][j] Hs[row][k]   (first column group only)
      if (grp == 0) {
#pragma unroll 4
        for (int mm = 0; mm < 32; ++mm) {
          const int t = 2 * mm + h;
          aW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[t * kLd + 32 * wr + r], Hs_s[t * kLd + 32 * wc + r], aW1, 0, 0, 0);
        }
      }
      // dZ[row][k] = (sum_j dnode[row][j] W1[j][k]) (1 - Hs[row][k]^2)
      {
        f32x16 acc = {0};
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const float4 a = *reinterpret_cast<const float4*>(&Ds[(32 * wr + r) * kLd + 8 * cc + 4 * h]);
          const float* wp = &Ws[(8 * cc + 4 * h) * kLd + 32 * wc + r];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[kLd], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * kLd], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * kLd], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float hv = Hs_s[row * kLd + col];
          Zs[row * kLd + col] = acc[reg] * (1.f - hv * hv);
        }
      }
      __syncthreads();                            // dZ complete; Ds / Hs_s are free: they take the feature chunks in turn
      if (more) AFB_ROWS(p0 + 64);
      // dW0[j][cg0 + 64 q + n] += sum_rows dZ[row][j] (feature row x mask)[row][n].  Chunk q + 1 goes to the other tile, and a wave can only
      // reach chunk q + 2's staging through the barrier of chunk q + 1, which every wave passes after its products on chunk q: one
      // barrier per chunk.
#pragma unroll
      for (int q = 0; q < 4; ++q) {               // unrolled: the accumulator of chunk q is a fixed register set
        if (q >= nch) break;
        float* Fs = (q & 1) ? Hs_s : Ds;
        const int col0 = cg0 + 64 * q + sc4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = srow + 16 * i;
          float4 e = rf[i];
          if (drop) {
            const uint32_t hr = rowh[row];
            e.x = lowbias32((uint32_t)(col0 + 0) ^ hr) >= thr ? e.x * keep_scale : 0.f;
            e.y = lowbias32((uint32_t)(col0 + 1) ^ hr) >= thr ? e.y * keep_scale : 0.f;
            e.z = lowbias32((uint32_t)(col0 + 2) ^ hr) >= thr ? e.z * keep_scale : 0.f;
            e.w = lowbias32((uint32_t)(col0 + 3) ^ hr) >= thr ? e.w * keep_scale : 0.f;
          }
          *reinterpret_cast<float4*>(&Fs[row * kLd + sc4]) = e;
        }
        __syncthreads();
        if (q + 1 < nch) AFB_GLOAD(q + 1);
        else if (more && with_r) AFB_RECON(p0 + 64);     // the staging registers are free: they carry the next step's recon rows
#pragma unroll 4
        for (int mm = 0; mm < 32; ++mm) {
          const int t = 2 * mm + h;
          aW0[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(Zs[t * kLd + 32 * wr + r], Fs[t * kLd + 32 * wc + r], aW0[q], 0, 0, 0);
        }
      }
#undef AFB_GLOAD
    }
#undef AFB_IDX
#undef AFB_ROWS
#undef AFB_RECON
    // ---- the window's sums into the gradient tensors ----
    if (grp == 0) {
      float* out = g.gW1 + (int64_t)c * 4096;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        atomicAdd(out + row * 64 + col, aW1[reg]);
      }
    }
    float* out0 = g.gW0 + (int64_t)64 * lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= nch) break;
      const int cn = cg0 + 64 * q + col;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (cn < n_c) atomicAdd(out0 + (int64_t)row * n_c + cn, aW0[q][reg]);
      }
    }
  }
}

}  // namespace

bool adj_fused_eligible(const matcha_shape& s, const matcha_frozen& f) {
  return s.mode == 1 && s.d == 64 && f.feat_row_pad == 64 && s.n_chrom >= 1 && s.n_chrom <= kMaxChrom && (options().disable_fused & 2) == 0;
}

int adj_fused_forward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o, const int64_t* ids, int64_t T,
                      const AdjWs& w, int r_chrom, bool save, float* node_out, float* x0, float* X, float* recon_out, hipStream_t st,
                      const int32_t* slot_map) {
  const int C = s.n_chrom;
  const bool train = o.training != 0 && o.p_drop_adj > 0.f;
  const bool recon = recon_out && r_chrom >= 0;
  AdjFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.ids = ids; a.order = w.order; a.seg = w.seg; a.bounds = f.bounds; a.feat_off = f.feat_off; a.feats = f.feats; a.feat_pad = f.feat_row_pad;
  a.w0 = p.adj_w0; a.w1 = p.adj_w1; a.attr = attr_src(f, s.n_attr); a.Wa = p.attr_w; a.ba = p.attr_b; a.Wn = p.next_w; a.bn = p.next_b;
  a.Hs = save ? w.Hs : nullptr; a.TH = recon ? w.TH : nullptr; a.node = node_out; a.x0 = x0; a.X = X; a.C = C;
  a.seed = o.seed; a.p_drop = train ? o.p_drop_adj : 0.f; a.slot_map = slot_map;
  const unsigned grid = (unsigned)(cdiv(T, 64) + C + 1);
  const size_t lds = (size_t)2 * kTile * sizeof(float);
  // expected feature columns of a uniformly drawn token (2 n_i d + 2 d^2 flop per token; + the attribute path and next_w here)
  double e_nc = 0.0;
  if (f.bounds_host && f.bounds_host[C] > 0) {
    for (int i = 0; i < C; ++i) { const double n_i = f.bounds_host[i + 1] - f.bounds_host[i]; e_nc += n_i * n_i; }
    e_nc /= (double)f.bounds_host[C];
  }
  {
    ProfScope ps(MATCHA_PROF_ADJ_ENCODE, (double)T * (2.0 * e_nc * 64 + 2.0 * 4096 + (X ? 2.0 * s.n_attr * 64 + 2.0 * 4096 : 0.0)), st);
    if (X) {
      MATCHA_TRY(check_attr(f, s.n_attr));
      auto k = adj_fused_fwd_kernel<true>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    } else {
      auto k = adj_fused_fwd_kernel<false>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    }
    MATCHA_CHECK_LAUNCH("adj_fused_fwd_kernel");
  }
  if (!recon_out) return MATCHA_OK;
  if (r_chrom < 0) {
    MATCHA_TRY(zero_async(recon_out, 2 * sizeof(float), st));
    return MATCHA_OK;
  }
  MATCHA_CHECK_ARG(p.recon_w && p.recon_b && f.inter && f.bounds_host, "adj_forward: recon tensors / bounds_host missing");
  const int32_t* r_dev = o.random_chrom_dev;      // the chromosome lives on the device: offsets and widths are read there, launches sized for the widest
  const int lo_r = r_dev ? 0 : f.bounds_host[r_chrom], n_r = r_dev ? s.max_bins : f.bounds_host[r_chrom + 1] - lo_r;
  AdjReconArgs b;
  memset(&b, 0, sizeof(b));
  b.ids = ids; b.order = w.order; b.seg = w.seg; b.counts = w.counts; b.TH = w.TH; b.Wr = p.recon_w + (int64_t)64 * lo_r; b.br = p.recon_b + lo_r;
  b.inter = f.inter; b.n_nodes = s.n_nodes; b.r = r_chrom; b.lo_r = lo_r; b.n_r = n_r; b.dnr = w.dTH; b.gW = w.rgrad; b.gb = w.rgrad + w.nr_pad * 64;
  b.slab = w.lossslab; b.r_dev = r_dev; b.bounds = f.bounds; b.C = s.n_chrom;
  int rgrid = (int)cdiv(T, 64);
  if (rgrid > 512) rgrid = 512;
  const size_t rlds = (size_t)3 * kTile * sizeof(float);
  // 2 d n_r flop per token outside chromosome r, three times that with the branch's backward in the same pass
  ProfScope ps(MATCHA_PROF_ADJ_RECON, (double)T * (1.0 - (double)n_r / (double)s.n_nodes) * 2.0 * 64 * n_r * (save ? 3.0 : 1.0), st);
  if (save) {                                     // (w.rgrad was zeroed by the sort's scan kernel)
    auto k = adj_recon_kernel<true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
    hipLaunchKernelGGL(k, dim3(rgrid), dim3(256), rlds, st, b);
  } else {
    auto k = adj_recon_kernel<false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
    hipLaunchKernelGGL(k, dim3(rgrid), dim3(256), rlds, st, b);
  }
  MATCHA_CHECK_LAUNCH("adj_recon_kernel");
  hipLaunchKernelGGL(adj_recon_sum_kernel, dim3(1), dim3(256), 0, st, w.lossslab, rgrid, w.counts, n_r, recon_out, r_dev, f.bounds, (int)s.n_chrom);
  MATCHA_CHECK_LAUNCH("adj_recon_sum_kernel");
  return MATCHA_OK;
}

int adj_fused_backward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o, const int64_t* ids, int64_t T,
                       const AdjWs& w, int r_chrom, const float* dX0, const float* drecon, matcha_tensors& g_, hipStream_t st,
                       const int32_t* slot_map, int32_t* touched) {
  const int C = s.n_chrom;
  const bool train = o.training != 0 && o.p_drop_adj > 0.f;
  const bool recon = r_chrom >= 0 && (drecon || o.beta != 0.f);
  AdjBwdArgs a;
  memset(&a, 0, sizeof(a));
  int n_apply = 0;
  if (recon) {
    MATCHA_CHECK_ARG(g_.recon_w && g_.recon_b && f.bounds_host, "adj_backward: recon gradient buffers missing");
    const int32_t* r_dev = o.random_chrom_dev;
    const int lo_r = r_dev ? 0 : f.bounds_host[r_chrom], n_r = r_dev ? s.max_bins : f.bounds_host[r_chrom + 1] - lo_r;
    a.rgW = w.rgrad; a.rgb = w.rgrad + w.nr_pad * 64; a.n_r = n_r; a.dWr = g_.recon_w + (int64_t)64 * lo_r; a.dbr = g_.recon_b + lo_r;
    n_apply = (int)cdiv((int64_t)n_r * 64, 256);
  }
  a.touched = touched; a.counts = w.counts;
  a.ids = ids; a.order = w.order; a.seg = w.seg; a.bounds = f.bounds; a.feat_off = f.feat_off; a.feats = f.feats; a.feat_pad = f.feat_row_pad;
  a.w1 = p.adj_w1; a.dX0 = dX0; a.dnr = recon ? w.dTH : nullptr; a.drecon = drecon; a.beta = o.beta; a.Hs = w.Hs; a.gW0 = g_.adj_w0; a.gW1 = g_.adj_w1;
  a.C = C; a.r = r_chrom; a.seed = o.seed; a.p_drop = train ? o.p_drop_adj : 0.f; a.slot_map = slot_map; a.r_dev = o.random_chrom_dev;
  // windows: ~2 per CU at large batches (every window ends in 64 x (64 + n_c) float atomics), one 64-row step per window at small ones
  const int64_t steps = cdiv(T, 64);
  int spi = (int)(steps / 640);
  spi = spi < 1 ? 1 : (spi > 16 ? 16 : spi);
  a.steps_per_item = spi;
  a.n_main = (int)(cdiv(steps, spi) + C);
  const unsigned grid = (unsigned)(a.n_main + 1 + n_apply);
  const size_t lds = (size_t)4 * kTile * sizeof(float);
  auto k = adj_fused_bwd_kernel;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  double e_nc = 0.0;
  if (f.bounds_host && f.bounds_host[C] > 0) {
    for (int i = 0; i < C; ++i) { const double n_i = f.bounds_host[i + 1] - f.bounds_host[i]; e_nc += n_i * n_i; }
    e_nc /= (double)f.bounds_host[C];
  }
  ProfScope ps(MATCHA_PROF_ADJ_BWD, (double)T * (2.0 * e_nc * 64 + 4.0 * 4096), st);      // dW0 (2 n_c d) + dW1 + dZ (2 d^2 each) per token
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
  MATCHA_CHECK_LAUNCH("adj_fused_bwd_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
