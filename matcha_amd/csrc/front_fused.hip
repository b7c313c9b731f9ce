// Fused front end of the encoder for embed_dim 64 (the metric's configuration), forward and backward.
//
// Forward (front_fwd_kernel): x0 = node_row + attribute_nn(attr_table[id])  (Modules.py:263-269),  X = tanh(next_w(x0))  (:270)
// in one token-major kernel: the embedding rows are gathered straight into an LDS tile (coalesced 256-B rows), the
// attribute Linear is a K = 32 MFMA GEMM on the gathered attribute rows, and the next_w GEMM consumes the x0 tile in
// place; x0 (needed by the weight gradient) and X are the only writes.
//
// Backward (front_bwd_kernel):
//   d x_hat = sum over the 8 per-head partials of fused_bwd.hip  ->  LayerNorm backward (no affine) + static-branch
//   gradient + tanh'  = dZ0            (Modules.py:519-521 backward, :270)
//   dX0 = dZ0 . Wn,  dWn += dZ0^T x0,  d bn += colsum(dZ0)                      (next_w, Modules.py:270)
//   dWa += dX0^T attr_table[id],  d ba += colsum(dX0)                           (attribute_nn, :263-264)
//   table front end: dtable[id] += dX0 (row 0 = padding_idx skipped);  adj front end: dX0 is handed to adj_backward.
// One token-major kernel instead of lnhat_bwd + 2 GEMM_TN (+ slab reduces) + GEMM_NN + scatter: per token it reads the
// 8 partials (2 KB), X, dXs, x0 (768 B), the id and its attribute row, and dZ0 / dX0 never reach HBM (table mode).
// Persistent workgroups walk 64-token tiles; next_w stays in LDS, the two weight gradients in MFMA accumulators; one slab
// per workgroup, summed in a fixed order by front_slab_reduce_kernel.  LDS 78 KB -> two workgroups per CU.
#include <string.h>

#include "attr_src.hpp"
#include "kernels.hpp"
#include "prep_heads.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kLd = 68;
constexpr int kTile = 64 * kLd;
constexpr int kLdA = 36;                          // attribute tile row stride: 32 columns + 4
constexpr int kAttrCols = 32;                     // attribute features handled by the fused kernel (n_attr <= 32, multiple of 4)
constexpr int kFrontSlab = 4096 + 64 * kAttrCols + 128;     // dWn [64][64] | dWa [64][32] | d bn [64] | d ba [64]
constexpr float kEps = 1e-5f;

struct FrontBwdArgs {
  const float* X; const float* dxh; int nslab; int64_t tcap; const float* dxpad; const float* dXs; const float* x0;
  const int64_t* ids; AttrSrc attr; int n_attr;
  const float* Wn;                                // next_w [64][64] ([out][in])
  const int32_t* count;                           // {Tr + 1, Tr, tiles}
  float* dX0;                                     // adj front end: [Tn, 64] output; table front end: null
  float* dtable;                                  // table front end: scatter target; adj: null
  float* slab;                                    // [gridDim.x][kFrontSlab]
};

__global__ __launch_bounds__(256, 2) void front_bwd_kernel(FrontBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ws = lds;                                // next_w, resident
  float* Zs = lds + 1 * kTile;                    // dZ0
  float* X0s = lds + 2 * kTile;                   // x0
  float* Ds = lds + 3 * kTile;                    // dX0
  float* As = lds + 4 * kTile;                    // attribute rows [64][kLdA]
  int* ids_s = reinterpret_cast<int*>(As + 64 * kLdA);   // [64] node id of each row (0: padding / past the end)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  const int T = g.count[0];                       // real tokens + the shared padding token
  const int ntiles = (T + 63) / 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<float4*>(&Ws[(srow + 16 * i) * kLd + sc4]) = *reinterpret_cast<const float4*>(g.Wn + (srow + 16 * i) * 64 + sc4);
  f32x16 aWn = {0}, aWa = {0};
  float csz = 0.f, csd = 0.f;

  // Register prefetch, one tile ahead: while tile t is computed, the rows of tile t + stride (X, the d x_hat sum, dXs, x0: four float4 per
  // staged row and thread, the ids and this thread's two attribute pieces) are in flight.  Loads are unconditional on clamped indices and
  // the masks apply to the values: no branch sits around a load.  (Without the prefetch every tile waited out its HBM round trips --
  // id -> attribute row even two dependent ones -- with only the CU's other workgroup to cover them: 0.33 of the HBM roof.)
  float4 pxv[4], pdv[4], psv[4], px0[4], pav[2];
  int pid = 0;
  const float4 dxp = *reinterpret_cast<const float4*>(g.dxpad + sc4);        // the shared padding token's d x_hat (fb_unfold2_kernel)
  const int a_row0 = tid >> 3, a_q = (tid & 7) * 4;                           // attribute pieces: rows a_row0 and a_row0 + 32, columns a_q .. a_q + 3
  const int a_qc = a_q < g.n_attr ? a_q : 0;
#define FBW_GLOAD(TILE)                                                                                  \
  do {                                                                                                   \
    const int64_t tb__ = (int64_t)(TILE) * 64;                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int64_t t__ = tb__ + srow + 16 * i__;                                                        \
      const int64_t tc__ = t__ < T ? t__ : (int64_t)T - 1;                                               \
      pxv[i__] = *reinterpret_cast<const float4*>(g.X + tc__ * 64 + sc4);                                \
      pdv[i__] = *reinterpret_cast<const float4*>(g.dxh + tc__ * 64 + sc4);                              \
      psv[i__] = *reinterpret_cast<const float4*>(g.dXs + tc__ * 64 + sc4);                              \
      px0[i__] = *reinterpret_cast<const float4*>(g.x0 + tc__ * 64 + sc4);                               \
    }                                                                                                    \
    {                                                                                                    \
      const int64_t t__ = tb__ + (tid & 63);                                                             \
      pid = (int)g.ids[t__ < T ? t__ : (int64_t)T - 1];                                                  \
    }                                                                                                    \
    _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__) {                                                \
      const int64_t t__ = tb__ + a_row0 + 32 * j__;                                                      \
      const int64_t id__ = g.ids[t__ < T ? t__ : (int64_t)T - 1];                                        \
      pav[j__] = *reinterpret_cast<const float4*>(g.attr.table + id__ * g.attr.ld + a_qc);               \
    }                                                                                                    \
  } while (0)
  if ((int)blockIdx.x < ntiles) FBW_GLOAD(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t t_base = (int64_t)tile * 64;
    __syncthreads();                              // previous tile's GEMMs are done with the working tiles
    // ---- stage: dZ0 (LayerNorm backward of the summed partials + static branch + tanh'), x0, ids, attribute rows ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      const int64_t t = t_base + row;
      const bool valid = t < T;
      const int64_t tc = valid ? t : (int64_t)T - 1;
      const float4 xv = pxv[i];
      float4 d = pdv[i];
      for (int hd = 1; hd < g.nslab; ++hd) {          // per-head slabs (deterministic / row-sparse modes): the other seven, added in order
        const float4 v = *reinterpret_cast<const float4*>(g.dxh + ((int64_t)hd * g.tcap + tc) * 64 + sc4);
        d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
      }
      if (tc >= T - 1) d = dxp;
      const float mean = group_sum16_dpp((xv.x + xv.y) + (xv.z + xv.w)) * (1.f / 64.f);
      const float a0 = xv.x - mean, a1 = xv.y - mean, a2 = xv.z - mean, a3 = xv.w - mean;
      const float rs = __builtin_amdgcn_rsqf(group_sum16_dpp((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) * (1.f / 64.f) + kEps);
      const float4 xh = make_float4(a0 * rs, a1 * rs, a2 * rs, a3 * rs);
      const float ma = group_sum16_dpp((d.x + d.y) + (d.z + d.w)) * (1.f / 64.f);
      const float mb = group_sum16_dpp((d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w)) * (1.f / 64.f);
      const float4 s = psv[i];
      const float m = valid ? 1.f : 0.f;
      float4 z;
      z.x = m * (rs * (d.x - ma - xh.x * mb) + s.x) * (1.f - xv.x * xv.x);
      z.y = m * (rs * (d.y - ma - xh.y * mb) + s.y) * (1.f - xv.y * xv.y);
      z.z = m * (rs * (d.z - ma - xh.z * mb) + s.z) * (1.f - xv.z * xv.z);
      z.w = m * (rs * (d.w - ma - xh.w * mb) + s.w) * (1.f - xv.w * xv.w);
      *reinterpret_cast<float4*>(&Zs[row * kLd + sc4]) = z;
      const float4 x0v = px0[i];
      *reinterpret_cast<float4*>(&X0s[row * kLd + sc4]) = make_float4(x0v.x * m, x0v.y * m, x0v.z * m, x0v.w * m);
    }
    if (tid < 64) ids_s[tid] = (t_base + tid < T) ? pid : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = a_row0 + 32 * j;
      const bool on = t_base + row < T && a_q < g.n_attr;
      const float4 v = pav[j];
      *reinterpret_cast<float4*>(&As[row * kLdA + a_q]) = on ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tile + (int)gridDim.x < ntiles) FBW_GLOAD(tile + (int)gridDim.x);      // next tile's rows: in flight during this tile's GEMMs
    __syncthreads();
    // ---- dX0 = dZ0 . Wn  (Wn stored [n][k]: column walk) ----
    {
      f32x16 acc = {0};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(&Zs[(32 * wr + r) * kLd + 8 * c + 4 * h]);
        const float* wp = &Ws[(8 * c + 4 * h) * kLd + 32 * wc + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[kLd], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * kLd], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * kLd], acc, 0, 0, 0);
      }
      const int col = 32 * wc + r;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        Ds[row * kLd + col] = acc[reg];
        if (!g.dX0) {
          const int id = ids_s[row];
          if (id != 0) atomicAdd(g.dtable + (int64_t)id * 64 + col, acc[reg]);      // embedding backward (padding_idx = 0 skipped)
        }
      }
    }
    // ---- dWn[n][k] += sum_t dZ0[t][n] x0[t][k];  d bn = column sums of dZ0 ----
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const int t = 2 * m + h;
      const float gz = Zs[t * kLd + 32 * wr + r];
      csz += gz;
      aWn = __builtin_amdgcn_mfma_f32_32x32x2f32(gz, X0s[t * kLd + 32 * wc + r], aWn, 0, 0, 0);
    }
    __syncthreads();                              // dX0 tile complete
    if (g.dX0) {
      // gradient rows for the adj front end / the sorted table gradient: whole 256-byte rows from the LDS tile (16 lanes x 16 B)
      // instead of 16 four-byte stores per lane in the accumulator layout
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;
        const int64_t t = t_base + row;
        if (t < T) *reinterpret_cast<float4*>(g.dX0 + t * 64 + sc4) = *reinterpret_cast<const float4*>(&Ds[row * kLd + sc4]);
      }
    }
    // ---- dWa[n][a] += sum_t dX0[t][n] attr[t][a]  (a < 32: the waves with wc = 0);  d ba = column sums of dX0 ----
    if (wc == 0) {
#pragma unroll 8
      for (int m = 0; m < 32; ++m) {
        const int t = 2 * m + h;
        const float gd = Ds[t * kLd + 32 * wr + r];
        csd += gd;
        aWa = __builtin_amdgcn_mfma_f32_32x32x2f32(gd, As[t * kLdA + r], aWa, 0, 0, 0);
      }
    }
  }
  // ---- workgroup slab ----
  float* slab = g.slab + (int64_t)blockIdx.x * kFrontSlab;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    slab[row * 64 + 32 * wc + r] = aWn[reg];
    if (wc == 0) slab[4096 + row * kAttrCols + r] = aWa[reg];
  }
  csz += __shfl_xor(csz, 32, 64);
  csd += __shfl_xor(csd, 32, 64);
  if (wc == 0 && h == 0) {
    slab[4096 + 64 * kAttrCols + 32 * wr + r] = csz;
    slab[4096 + 64 * kAttrCols + 64 + 32 * wr + r] = csd;
  }
}

struct FrontFwdArgs {
  const int64_t* ids; const float* table; const float* dense;       // node rows: table[id] (table front end) or dense[t] (adj)
  AttrSrc attr; int n_attr;
  const float* Wa; const float* ba; const float* Wn; const float* bn;
  const int32_t* count;
  float* x0; float* X;
  int nprep; PrepArgs prep;      // the first nprep blocks build the step's weight forms instead (prep_heads.hpp): they depend on the parameters only
  int prep_walks;                // ... and then walk tiles like the other blocks (the launch has one block per slot of the chip) or leave
};

__global__ __launch_bounds__(256, 3) void front_fwd_kernel(FrontFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // Block roles.  The per-step weight forms of the encoder kernels (B_h, M_h, the fragment stream) depend on the parameters only, this kernel on
  // the batch only: their 72 blocks ride in this launch instead of being a launch of their own between the front end and the encoder (11 us
  // of every step).  Large batches: the launch still has ONE block per slot of the chip -- 72 blocks on top of a full grid start the last
  // front-end blocks late and stretch the kernel by what they took --, so the role blocks build weight forms and THEN walk tiles like the
  // others, and the last (partial) round of tiles goes to the blocks without a role first.
  const int nprep = g.nprep;
  if ((int)blockIdx.x < nprep) {
    const int b = blockIdx.x;
    prep_heads_role(g.prep, b % kPrepGridX, (b / kPrepGridX) % kPrepGridY, b / (kPrepGridX * kPrepGridY), lds);
    if (!g.prep_walks) return;
    __syncthreads();
  }
  const int nlate = g.prep_walks ? nprep : 0;           // front-end blocks that start late
  const int vb = (int)blockIdx.x - (nprep - nlate), Gf = (int)gridDim.x - (nprep - nlate);
  // The two weight matrices live in REGISTERS as MFMA operand fragments (32 + 16 per lane) instead of LDS tiles: 44 KB of LDS per
  // workgroup instead of 70, i.e. three workgroups per CU gathering rows -- this kernel is a gather, bytes in flight are what it needs
  float* Es = lds;                                // node rows, then x0 in place
  float* As = lds + kTile;                        // attribute rows [64][kLdA]
  float* Xs = As + 64 * kLdA;                     // X tile on its way out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  const int T = g.count[0];
  const int ntiles = (T + 63) / 64;
  // tiles of this block: vb + it Gf in the full rounds it < q; the r tiles of the last round go to the blocks vb = nlate, nlate + 1, ... (mod Gf)
  const int q = ntiles / Gf, r_last = ntiles - q * Gf;
  const int last_slot = vb >= nlate ? vb - nlate : vb - nlate + Gf;
  const int tile_last = last_slot < r_last ? q * Gf + last_slot : 0x3FFFFFF;
#define FF_TILE(IT) ((IT) < q ? vb + (IT) * Gf : ((IT) == q ? tile_last : 0x3FFFFFF))
  float4 wn[8], wa[kAttrCols / 8];                // B fragments: next_w[32 wc + r][8 c + 4 h ..], attribute_nn.weight[32 wc + r][8 c + 4 h ..]
#pragma unroll
  for (int c = 0; c < 8; ++c) wn[c] = *reinterpret_cast<const float4*>(g.Wn + (32 * wc + r) * 64 + 8 * c + 4 * h);
#pragma unroll
  for (int c = 0; c < kAttrCols / 8; ++c) {
    const int q = 8 * c + 4 * h;
    wa[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < g.n_attr) wa[c] = *reinterpret_cast<const float4*>(g.Wa + (32 * wc + r) * g.n_attr + q);
  }
  const int col = 32 * wc + r;
  const float bav = g.ba[col], bnv = g.bn[col];

  // Two-stage prefetch, one tile ahead for the rows and two for the ids: while tile t is computed, the node / attribute rows of tile
  // t + stride are in flight (their ids arrived during tile t - stride).  Without it every tile waited out id -> row, two dependent
  // HBM round trips, with only the CU's other workgroup to cover them: 13 % of the HBM roof on a 4 GiB table.
  const int arow0 = tid >> 3, aq = (tid & 7) * 4;     // attribute pieces: rows arow0 and arow0 + 32, columns aq .. aq + 3
  int64_t id_e[4], id_a[2];
  float4 pe[4], pa[2], qe[4], qa[2];                  // rows of the next tile (pe, pa) and of the one after (qe, qa)
  // No predicated loads (hipcc turns `cond ? load : 0` into a branch + s_waitcnt vmcnt(0) per load, i.e. one load in flight): token
  // indices are clamped to the last token instead -- rows past the end hold a copy of it and are never stored.
  const float amask = aq < g.n_attr ? 1.f : 0.f;
  const int aqc = aq < g.n_attr ? aq : g.n_attr - 4;
  const int64_t t_last = (int64_t)T - 1;
#define FF_IDS_GLOAD(TILE)                                                                               \
  do {                                                                                                   \
    const int64_t tb__ = (int64_t)(TILE) * 64;                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int64_t t__ = tb__ + srow + 16 * i__;                                                        \
      id_e[i__] = g.ids[t__ < t_last ? t__ : t_last];                                                    \
    }                                                                                                    \
    _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__) {                                                \
      const int64_t t__ = tb__ + arow0 + 32 * j__;                                                       \
      id_a[j__] = g.ids[t__ < t_last ? t__ : t_last];                                                    \
    }                                                                                                    \
  } while (0)
#define FF_ROWS_GLOAD(TILE)                                                                              \
  do {                                                                                                   \
    const int64_t tb__ = (int64_t)(TILE) * 64;                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int64_t t__ = tb__ + srow + 16 * i__;                                                        \
      const float* src__ = g.table ? g.table + id_e[i__] * 64 : g.dense + (t__ < t_last ? t__ : t_last) * 64; \
      pe[i__] = *reinterpret_cast<const float4*>(src__ + sc4);                                           \
    }                                                                                                    \
    _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__) {                                                \
      const float4 v__ = *reinterpret_cast<const float4*>(g.attr.table + id_a[j__] * g.attr.ld + aqc);   \
      pa[j__] = make_float4(v__.x * amask, v__.y * amask, v__.z * amask, v__.w * amask);                 \
    }                                                                                                    \
  } while (0)
  {
    FF_IDS_GLOAD(FF_TILE(0));
    FF_ROWS_GLOAD(FF_TILE(0));
#pragma unroll
    for (int i = 0; i < 4; ++i) qe[i] = pe[i];
    qa[0] = pa[0]; qa[1] = pa[1];
    // (qe, qa) = the first tile; now the second into (pe, pa) ... rotated below so that (pe, pa) is always the tile about to be used
    FF_IDS_GLOAD(FF_TILE(1));
    FF_ROWS_GLOAD(FF_TILE(1));
    FF_IDS_GLOAD(FF_TILE(2));
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = pe[i]; pe[i] = qe[i]; qe[i] = t; }
    { const float4 t0 = pa[0], t1 = pa[1]; pa[0] = qa[0]; pa[1] = qa[1]; qa[0] = t0; qa[1] = t1; }
  }
  for (int it = 0;; ++it) {
    const int tile = FF_TILE(it);
    if (tile >= ntiles) break;
    const int64_t t_base = (int64_t)tile * 64;
    __syncthreads();
    // ---- the prefetched node rows (16 lanes x float4 per 256-B row) and attribute rows -> LDS; next tile's rows, the one after's ids ----
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Es[(srow + 16 * i) * kLd + sc4]) = pe[i];
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&As[(arow0 + 32 * j) * kLdA + aq]) = pa[j];
    // (pe, pa) <- the tile after this one (in flight since the previous trip); its registers take the loads of the tile two ahead
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = qe[i]; pe[i] = t; }
    pa[0] = qa[0]; pa[1] = qa[1];
    {
      float4 se[4], sa[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) se[i] = pe[i];
      sa[0] = pa[0]; sa[1] = pa[1];
      FF_ROWS_GLOAD(FF_TILE(it + 2));                 // into (pe, pa) ...
#pragma unroll
      for (int i = 0; i < 4; ++i) { qe[i] = pe[i]; pe[i] = se[i]; }     // ... which become (qe, qa); (pe, pa) = the next tile again
      qa[0] = pa[0]; qa[1] = pa[1]; pa[0] = sa[0]; pa[1] = sa[1];
    }
    FF_IDS_GLOAD(FF_TILE(it + 3));
    __syncthreads();
    // ---- x0 = node_row + attr . Wa^T + ba   (K = 32) ----
    {
      f32x16 acc = {0};
#pragma unroll
      for (int c = 0; c < kAttrCols / 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLdA + 8 * c + 4 * h]);
        const float4 b = wa[c];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const float v = acc[reg] + bav + Es[row * kLd + col];
        Es[row * kLd + col] = v;                                      // in place: this lane owns the element
      }
    }
    __syncthreads();
    // x0 and (below) X leave as whole 256-byte rows read back from their LDS tiles -- 16 lanes x 16 bytes -- not as 16 four-byte
    // stores per lane in the accumulator layout: this kernel only moves bytes, and store ISSUE was what bounded it
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      if (g.x0 && t_base + row < T) *reinterpret_cast<float4*>(g.x0 + (t_base + row) * 64 + sc4) = *reinterpret_cast<const float4*>(&Es[row * kLd + sc4]);
    }
    // ---- X = tanh(x0 . Wn^T + bn) ----
    {
      f32x16 acc = {0};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(&Es[(32 * wr + r) * kLd + 8 * c + 4 * h]);
        const float4 b = wn[c];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        Xs[row * kLd + col] = fast_tanh(acc[reg] + bnv);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      if (t_base + row < T) *reinterpret_cast<float4*>(g.X + (t_base + row) * 64 + sc4) = *reinterpret_cast<const float4*>(&Xs[row * kLd + sc4]);
    }
  }
}

// ---- round 6: the forward for attribute tables with get_attributes' structure (attr_mode 1; main.py:497-512) -------------------------------
// A row of that table is one-hot(chromosome) || coordinate, so attribute_nn(row) = Wa[:, chrom] + coord * Wa[:, C] + ba: TWO fused multiply-adds
// per output, not a K = 32 product -- and nothing to gather (one random row per token instead of two).  What is left of the
// kernel's arithmetic, next_w (K = 64), runs on the bf16 matrix pipe as fp32-accurate plane products (bf16x3.hpp): the weights are split once
// per workgroup into register fragments, the x0 tile is split ONCE by the thread that forms it (in the staging layout: 16 lanes x float4 per
// row, where x0 = gathered row + the two attribute terms costs 12 vector instructions) and kept in LDS as planes only; x0 leaves for HBM from
// the staging registers, X from the accumulators (16 bytes per lane: four lanes cover 64 contiguous bytes of a row).  Per 64-token tile: 48
// bf16 MFMAs of 16 cycles on the matrix pipe instead of 48 f32 MFMAs of 64 cycles on the vector ALUs, two barriers instead of four, 40 KB of
// LDS.  The in-step gather on a 4 GiB table: 0.22 -> see DESIGN.md 4.5 (the f32 products were 40 % of the kernel's time).
constexpr int kFPS = 80;                 // bf16 per plane row (40 dwords: the 16-byte row-fragment reads are conflict-free, fused_bwd.hip)
constexpr int kFPlane = 64 * kFPS;       // bf16 per plane of a 64-token tile
constexpr int kFwd2LdsFloats = (3 * kFPlane) / 2 + kAttrCols * 64 + 256 + 64;
// in-situ ablations (development builds, wrong results on purpose): -DFF2_ABL=<bits>  1: no decode, 2: no X stores, 4: no MFMAs, 8: no operand split, 16: no tanh
#ifndef FF2_ABL
#define FF2_ABL 0
#endif
// the rows this kernel writes (X; x0 in a training forward) are read by another kernel later: -DFF2_NT=1 streams them past L2 (non-temporal),
// as gather_rows_kernel does -- A/B'd on the 4 GiB table and in the training step
#ifndef FF2_NT
#define FF2_NT 0
#endif
#if FF2_NT
#define FF2_STORE4(P, A, B, C, D) __builtin_nontemporal_store((f32x4){(A), (B), (C), (D)}, reinterpret_cast<f32x4*>(P))
#else
#define FF2_STORE4(P, A, B, C, D) (*reinterpret_cast<f32x4*>(P) = (f32x4){(A), (B), (C), (D)})
#endif

__device__ __forceinline__ Frag3 frag_row64(const short* __restrict__ p) {
  Frag3 f;
  f.h = *reinterpret_cast<const u32x4*>(p); f.m = *reinterpret_cast<const u32x4*>(p + kFPlane); f.l = *reinterpret_cast<const u32x4*>(p + 2 * kFPlane);
  return f;
}

__global__ __launch_bounds__(256, 3) void front_fwd2_kernel(FrontFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nprep = g.nprep;
  if ((int)blockIdx.x < nprep) {                        // block roles: the step's weight forms first (see front_fwd_kernel)
    const int b = blockIdx.x;
    prep_heads_role(g.prep, b % kPrepGridX, (b / kPrepGridX) % kPrepGridY, b / (kPrepGridX * kPrepGridY), lds);
    if (!g.prep_walks) return;
    __syncthreads();
  }
  const int nlate = g.prep_walks ? nprep : 0;
  const int vb = (int)blockIdx.x - (nprep - nlate), Gf = (int)gridDim.x - (nprep - nlate);
  short* Xp = reinterpret_cast<short*>(lds);                          // x0 as three bf16 planes [3][64][kFPS]
  float* WaT = lds + (3 * kFPlane) / 2;                               // [n_attr][64]: row a < n_attr - 1 = attribute_nn.weight[:, a]; row n_attr - 1 = zeros
  int* cinfo = reinterpret_cast<int*>(WaT + kAttrCols * 64);          // [2][64] x (row of WaT, coordinate): double-buffered, decoded one tile ahead
  int* bnd = cinfo + 256;                                             // [n_attr] chromosome bounds 0, n_0, n_0 + n_1, ..., N
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kq = lane >> 4;
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  const int T = g.count[0];
  const int ntiles = (T + 63) / 64;
  const int q = ntiles / Gf, r_last = ntiles - q * Gf;
  const int last_slot = vb >= nlate ? vb - nlate : vb - nlate + Gf;
  const int tile_last = last_slot < r_last ? q * Gf + last_slot : 0x3FFFFFF;
  const int na = g.n_attr;
  for (int i = tid; i < na * 64; i += 256) {
    const int a = i >> 6, c = i & 63;
    WaT[i] = a < na - 1 ? g.Wa[c * na + a] : 0.f;
  }
  if (tid < na) bnd[tid] = g.attr.bounds[tid];
  // next_w as the A operand of X^T = Wn x0^T: wave w owns output features 16 w + {0..15}; lane (c16, kq), step s: Wn[16 w + c16][32 s + 8 kq + {0..7}]
  Frag3 Wf[2];
  {
    const float* wp = g.Wn + (16 * wave + c16) * 64 + 8 * kq;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = wp[32 * s2 + j];
      Wf[s2] = split8(v);
    }
  }
  const float4 bn4 = *reinterpret_cast<const float4*>(g.bn + 16 * wave + 4 * kq);
  const float4 ba4 = *reinterpret_cast<const float4*>(g.ba + sc4);
  const float4 wl4 = make_float4(g.Wa[(sc4 + 0) * na + na - 1], g.Wa[(sc4 + 1) * na + na - 1], g.Wa[(sc4 + 2) * na + na - 1], g.Wa[(sc4 + 3) * na + na - 1]);
  const float scale = g.attr.scale;

  int64_t id_e[4];
  float4 pe[4], qe[4];
  const int64_t t_last = (int64_t)T - 1;
#define FF2_IDS_GLOAD(TILE)                                                                              \
  do {                                                                                                   \
    const int64_t tb__ = (int64_t)(TILE) * 64;                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int64_t t__ = tb__ + srow + 16 * i__;                                                        \
      id_e[i__] = g.ids[t__ < t_last ? t__ : t_last];                                                    \
    }                                                                                                    \
  } while (0)
#define FF2_ROWS_GLOAD(TILE)                                                                             \
  do {                                                                                                   \
    const int64_t tb__ = (int64_t)(TILE) * 64;                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                \
      const int64_t t__ = tb__ + srow + 16 * i__;                                                        \
      const float* src__ = g.table ? g.table + id_e[i__] * 64 : g.dense + (t__ < t_last ? t__ : t_last) * 64; \
      pe[i__] = *reinterpret_cast<const float4*>(src__ + sc4);                                           \
    }                                                                                                    \
  } while (0)
  // token `lane` of a tile -> (row of WaT, coordinate): a search over the <= 31 bounds in LDS and one correctly rounded division (attr_src.hpp)
#define FF2_ID_OF(TILE) g.ids[((int64_t)(TILE) * 64 + lane) < t_last ? ((int64_t)(TILE) * 64 + lane) : t_last]
#define FF2_DECODE(ID, BUF)                                                                              \
  do {                                                                                                   \
    const int nb__ = na - 1;                                                                             \
    const int id__ = (int)(ID);                                                                          \
    const int top__ = bnd[nb__];                                                                         \
    const int idc__ = id__ < 1 ? 1 : (id__ > top__ ? top__ : id__);                                      \
    const int c__ = attr_chrom(bnd, nb__, idc__);                                                        \
    const bool real__ = id__ >= 1 && id__ <= top__;                                                      \
    cinfo[(BUF) * 128 + 2 * lane] = real__ ? c__ : nb__;                                                 \
    cinfo[(BUF) * 128 + 2 * lane + 1] = __float_as_int(real__ ? __fdiv_rn((float)(idc__ - bnd[c__] - 1), scale) : 0.f); \
  } while (0)
  int64_t idc;
  {
    // prologue: the ids of the first three tiles in ONE round trip, the rows of the first two in the next (the loop's steady state has the
    // rows of two tiles and the ids of a third in flight; issuing ids -> rows -> ids -> rows here was five dependent HBM round trips,
    // a fifth of the kernel at 65 536 rows)
    int64_t ia[4], ib[4];
    FF2_IDS_GLOAD(FF_TILE(0));
#pragma unroll
    for (int i = 0; i < 4; ++i) ia[i] = id_e[i];
    FF2_IDS_GLOAD(FF_TILE(1));
#pragma unroll
    for (int i = 0; i < 4; ++i) ib[i] = id_e[i];
    FF2_IDS_GLOAD(FF_TILE(2));
    const int64_t id0 = FF2_ID_OF(FF_TILE(0));
    idc = FF2_ID_OF(FF_TILE(1));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t t0__ = (int64_t)FF_TILE(0) * 64 + srow + 16 * i, t1__ = (int64_t)FF_TILE(1) * 64 + srow + 16 * i;
      const float* s0 = g.table ? g.table + ia[i] * 64 : g.dense + (t0__ < t_last ? t0__ : t_last) * 64;
      const float* s1 = g.table ? g.table + ib[i] * 64 : g.dense + (t1__ < t_last ? t1__ : t_last) * 64;
      pe[i] = *reinterpret_cast<const float4*>(s0 + sc4);
      qe[i] = *reinterpret_cast<const float4*>(s1 + sc4);
    }
    __syncthreads();                                   // bnd is staged
    if (wave == 0) FF2_DECODE(id0, 0);
  }
  for (int it = 0;; ++it) {
    const int tile = FF_TILE(it);
    if (tile >= ntiles) break;
    const int64_t t_base = (int64_t)tile * 64;
    __syncthreads();                                   // the previous tile's fragment reads are done; this tile's (row, coordinate) pairs are visible
    // ---- x0 = node row + Wa[:, chrom] + coord Wa[:, C] + ba in the staging layout -> HBM (training) and, split into planes, -> LDS ----
    {
      const int* ci = cinfo + (it & 1) * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;
        const int wr_ = ci[2 * row];
        const float co = __int_as_float(ci[2 * row + 1]);
        const float4 wc_ = *reinterpret_cast<const float4*>(&WaT[wr_ * 64 + sc4]);
        const float4 e = pe[i];
        float4 v;
        v.x = (__builtin_fmaf(co, wl4.x, wc_.x) + ba4.x) + e.x; v.y = (__builtin_fmaf(co, wl4.y, wc_.y) + ba4.y) + e.y;
        v.z = (__builtin_fmaf(co, wl4.z, wc_.z) + ba4.z) + e.z; v.w = (__builtin_fmaf(co, wl4.w, wc_.w) + ba4.w) + e.w;
        if (g.x0 && t_base + row < T) FF2_STORE4(g.x0 + (t_base + row) * 64 + sc4, v.x, v.y, v.z, v.w);
        P3 p0, p1;
        if (FF2_ABL & 8) { p0 = P3{__float_as_uint(v.x), __float_as_uint(v.y), 0u}; p1 = P3{__float_as_uint(v.z), __float_as_uint(v.w), 0u}; }
        else { p0 = split2(v.x, v.y); p1 = split2(v.z, v.w); }
        short* d = Xp + row * kFPS + sc4;
        *reinterpret_cast<u32x2*>(d) = (u32x2){p0.h, p1.h}; *reinterpret_cast<u32x2*>(d + kFPlane) = (u32x2){p0.m, p1.m};
        *reinterpret_cast<u32x2*>(d + 2 * kFPlane) = (u32x2){p0.l, p1.l};
      }
    }
    // (pe) <- the tile after this one (in flight since the previous trip); its registers take the loads of the tile two ahead
    {
      float4 se[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) se[i] = qe[i];
      FF2_ROWS_GLOAD(FF_TILE(it + 2));
#pragma unroll
      for (int i = 0; i < 4; ++i) { qe[i] = pe[i]; pe[i] = se[i]; }
    }
    FF2_IDS_GLOAD(FF_TILE(it + 3));
    __syncthreads();
    if (wave == 0 && !(FF2_ABL & 1)) FF2_DECODE(idc, (it + 1) & 1);      // the next tile's pairs (its ids arrived during the previous tile)
    idc = FF2_ID_OF(FF_TILE(it + 2));
    // ---- X^T = tanh(Wn x0^T + bn): lane (c16, kq) ends with token 16 tb + c16 and features 16 wave + 4 kq + {0..3} ----
    {
      const short* bp = Xp + c16 * kFPS + 8 * kq;
      f32x4 acc[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) acc[tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          const Frag3 b = frag_row64(bp + tb * 16 * kFPS + 32 * s2);
          if (FF2_ABL & 4) { acc[tb][0] += __uint_as_float(b.h[0] ^ b.m[1] ^ b.l[2]); continue; }
          acc[tb] = mma6(acc[tb], Wf[s2], b);
        }
      }
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const int64_t t = t_base + 16 * tb + c16;
        f32x4 o;
        if (FF2_ABL & 16) { o[0] = acc[tb][0] + bn4.x; o[1] = acc[tb][1] + bn4.y; o[2] = acc[tb][2] + bn4.z; o[3] = acc[tb][3] + bn4.w; }
        else { o[0] = fast_tanh(acc[tb][0] + bn4.x); o[1] = fast_tanh(acc[tb][1] + bn4.y); o[2] = fast_tanh(acc[tb][2] + bn4.z); o[3] = fast_tanh(acc[tb][3] + bn4.w); }
        if (t < T && (!(FF2_ABL & 2) || o[0] == 12345.f)) FF2_STORE4(g.X + t * 64 + 16 * wave + 4 * kq, o[0], o[1], o[2], o[3]);
      }
    }
  }
#undef FF2_IDS_GLOAD
#undef FF2_ROWS_GLOAD
#undef FF2_ID_OF
#undef FF2_DECODE
}

struct FrontReduceArgs {
  const float* slab; int nwg; int n_attr;
  float* dWn; float* dWa; float* dbn; float* dba;
  int32_t* touched;   // table front end: the two "this group of parameters received a gradient" flags, set here instead of by a launch of their own; null: not asked
};
// fixed-order sum over the workgroup slabs (64 outputs x 16 lanes per block), accumulated into the gradient tensors
__global__ __launch_bounds__(1024) void front_slab_reduce_kernel(FrontReduceArgs a) {
  __shared__ float part[16][64];
  if (a.touched && blockIdx.x == 0 && threadIdx.x < 2) a.touched[threadIdx.x] = 1;
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < kFrontSlab) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // four slabs in flight per thread (one at a time was a chain of nwg / 16 round trips)
    int b = q;
    for (; b + 48 < a.nwg; b += 64) {
      s0 += a.slab[(int64_t)b * kFrontSlab + i]; s1 += a.slab[(int64_t)(b + 16) * kFrontSlab + i];
      s2 += a.slab[(int64_t)(b + 32) * kFrontSlab + i]; s3 += a.slab[(int64_t)(b + 48) * kFrontSlab + i];
    }
    for (; b < a.nwg; b += 16) s0 += a.slab[(int64_t)b * kFrontSlab + i];
    s = (s0 + s1) + (s2 + s3);
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0 && i < kFrontSlab) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += part[u][o];
    if (i < 4096) a.dWn[i] += t;
    else if (i < 4096 + 64 * kAttrCols) {
      const int n = (i - 4096) / kAttrCols, c = (i - 4096) % kAttrCols;
      if (c < a.n_attr) a.dWa[n * a.n_attr + c] += t;
    } else if (i < 4096 + 64 * kAttrCols + 64) a.dbn[i - 4096 - 64 * kAttrCols] += t;
    else a.dba[i - 4096 - 64 * kAttrCols - 64] += t;
  }
}

int front_grid() {
  const int n = 2 * device_cu_count();
  return n > 1024 ? 1024 : n;
}

}  // namespace

int launch_front_fwd(const matcha_tensors& p, const int64_t* ids, const float* table, const float* dense, const matcha_frozen& f, int n_attr,
                     const Ragged& rg, int64_t tcap, float* x0, float* X, hipStream_t st, const PrepSpec* prep) {
  // the fused front end GATHERS attribute rows (rows padded to one 128-byte unit when the caller padded them, attr_ld): its row pieces are
  // loaded by eight threads per row inside a register prefetch pipeline; rebuilding them from the node id there (attr_mode 1) put a branch
  // around the loads and cost 20 % of front_bwd_kernel -- embed_fwd_kernel and the fused adj forward, one thread / one lane pair per row, do rebuild
  MATCHA_CHECK_ARG(f.attr_table, "front end: attr_table is required (also under attr_mode 1)");
  FrontFwdArgs g;
  // attr_mode 1 (the table has get_attributes' structure): front_fwd2_kernel -- no attribute rows gathered, no attribute product; any other
  // table: front_fwd_kernel gathers its rows (padded to one 128-byte unit when the caller padded them) for the K = 32 product
  const bool computed = f.attr_mode == 1 && f.attr_bounds && n_attr <= kAttrCols;
  g.ids = ids; g.table = table; g.dense = dense; g.attr = computed ? attr_src(f, n_attr) : attr_src_table_first(f, n_attr); g.n_attr = n_attr;
  g.Wa = p.attr_w; g.ba = p.attr_b; g.Wn = p.next_w; g.bn = p.next_b; g.count = rg.count; g.x0 = x0; g.X = X;
  const int slots = front_grid() / 2 * 3;               // three workgroups per CU (44 / 40 KB of LDS each)
  const int64_t max_tiles = cdiv(tcap, 64);
  const size_t lds = computed ? (size_t)kFwd2LdsFloats * sizeof(float) : ((size_t)2 * kTile + 64 * kLdA) * sizeof(float);
  static_assert(((size_t)2 * kTile + 64 * kLdA) >= (size_t)kPrepLdsFloats && kFwd2LdsFloats >= kPrepLdsFloats, "the weight-form role needs its LDS inside the front end's");
  g.nprep = 0;
  if (prep) {
    prep_heads_args(*prep->p, prep->folded, prep->merged, prep->frag, g.prep);
    g.nprep = kPrepBlocks;
  } else {
    memset(&g.prep, 0, sizeof(g.prep));
  }
  // large batches: one block per slot of the chip, the role blocks among them; small ones: the role blocks + one block per tile
  g.prep_walks = (g.nprep > 0 && max_tiles + g.nprep > slots) ? 1 : 0;      // (measured against 72 blocks on top of a full grid: 5 us per 65 536-row step)
  int64_t grid = g.prep_walks ? slots : g.nprep + (max_tiles < slots ? max_tiles : slots);
  if (grid < g.nprep + 1) grid = g.nprep + 1;
  // algorithmic bytes per token: id 8 + node row 256 + attribute row read (attr_mode 1: rebuilt from the id, nothing read); x0 and X rows
  // written (x0 == null: an inference forward -- nobody reads the pre-activation rows, they are not written)
  ProfScope ps(MATCHA_PROF_FRONT_FWD, (double)tcap * (8.0 + 256.0 + (computed ? 0.0 : 4.0 * n_attr) + (x0 ? 512.0 : 256.0)), st);
  if (computed) {
    hipLaunchKernelGGL(front_fwd2_kernel, dim3((unsigned)grid), dim3(256), lds, st, g);
    MATCHA_CHECK_LAUNCH("front_fwd2_kernel");
  } else {
    hipLaunchKernelGGL(front_fwd_kernel, dim3((unsigned)grid), dim3(256), lds, st, g);
    MATCHA_CHECK_LAUNCH("front_fwd_kernel");
  }
  return MATCHA_OK;
}

bool front_bwd_supported(int d, int n_attr) { return d == 64 && n_attr >= 4 && n_attr <= kAttrCols && n_attr % 4 == 0; }
size_t front_bwd_ws_floats() { return (size_t)1024 * kFrontSlab; }

int launch_front_bwd(const matcha_tensors& p, const float* X, const float* dxh, int nslab, int64_t tcap, const float* dxpad, const float* dXs, const float* x0,
                     const int64_t* ids, const matcha_frozen& f, int n_attr, const Ragged& rg, float* dX0, float* dtable, float* ws,
                     matcha_tensors& grads, hipStream_t st, int32_t* touched) {
  MATCHA_CHECK_ARG(f.attr_table, "front end: attr_table is required (also under attr_mode 1)");
  FrontBwdArgs g;
  g.X = X; g.dxh = dxh; g.nslab = nslab; g.tcap = tcap; g.dxpad = dxpad; g.dXs = dXs; g.x0 = x0; g.ids = ids; g.attr = attr_src_table_first(f, n_attr); g.n_attr = n_attr;
  g.Wn = p.next_w; g.count = rg.count; g.dX0 = dX0; g.dtable = dtable; g.slab = ws;
  int grid = front_grid();
  const int64_t max_tiles = cdiv(tcap, 64);
  if (grid > max_tiles) grid = (int)max_tiles;
  const size_t lds = ((size_t)4 * kTile + 64 * kLdA + 64) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(front_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  {
    // algorithmic bytes per token: the d x_hat partials (8 per-head slabs, or 1 when the heads were added with atomics) + X + dXs + x0, id,
    // attribute row; table mode adds 256 B of atomics
    ProfScope ps(MATCHA_PROF_FRONT_BWD, (double)tcap * ((3.0 + nslab) * 256.0 + 8.0 + (g.attr.mode == 1 ? 0.0 : 4.0 * n_attr) + 256.0), st);
    hipLaunchKernelGGL(front_bwd_kernel, dim3(grid), dim3(256), lds, st, g);
    MATCHA_CHECK_LAUNCH("front_bwd_kernel");
  }
  FrontReduceArgs a;
  a.slab = ws; a.nwg = grid; a.n_attr = n_attr; a.dWn = grads.next_w; a.dWa = grads.attr_w; a.dbn = grads.next_b; a.dba = grads.attr_b;
  a.touched = touched;
  hipLaunchKernelGGL(front_slab_reduce_kernel, dim3((unsigned)cdiv(kFrontSlab, 64)), dim3(1024), 0, st, a);
  MATCHA_CHECK_LAUNCH("front_slab_reduce_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
