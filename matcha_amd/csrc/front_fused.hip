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

// ---- round 6: the forward for attribute tables with get_attributes' structure (attr_mode 1; main.py:497-512): front_fwd3_kernel --------------
// A row of that table is one-hot(chromosome) || coordinate, so attribute_nn(row) = Wa[:, chrom] + coord * Wa[:, C] + ba: TWO fused multiply-adds
// per output, not a K = 32 product -- and nothing to gather (one random row per token instead of two).  What is left of the
// kernel's arithmetic, next_w (K = 64), runs on the bf16 matrix pipe as fp32-accurate plane products (bf16x3.hpp).
// WAVE-INDEPENDENT (no barrier, no staged tile): one wavefront = 16 tokens per trip, its own loop.  Lane (c16, kq) gathers 64 bytes of token
// c16's row -- features 32 s + 8 kq + {0..7}, s = 0, 1: exactly the eight contraction slots per step that v_mfma_f32_16x16x32_bf16 wants from
// this lane as the B operand of X^T = Wn x0^T -- adds the two attribute terms from LDS, splits in registers, and multiplies against next_w's
// planes read from LDS as A fragments (shared by all wavefronts, staged once per workgroup).  Nothing of the batch goes through LDS and
// nothing synchronises: latency is hidden the way gather_rows_kernel hides it, by sixteen wavefronts per CU each with two trips of rows in
// flight.  Two workgroup-tiled versions were built and measured first (64-token tiles, x0 planes in LDS, two barriers per tile; in the
// history at `front_fwd2_kernel`): with two register row buffers 47-50 us on the 4 GiB table, and with a THREE-deep register prefetch no
// better -- s_waitcnt vmcnt counts in order and hipcc's loop analysis merges the pending-load state at the loop header conservatively, so
// every trip waited (vmcnt(8)) for the rows issued one trip earlier whatever the source said; a 64-bit id load whose upper half nobody read
// even drew a vmcnt(0) into the middle of a tile (the allocator reused the dead half).  This version: 46 us there, 34-35 us in the step.
constexpr int kFPS = 80;                 // bf16 per plane row of next_w (40 dwords: the 16-byte row-fragment reads are conflict-free, fused_bwd.hip)
#define FF2_STORE4(P, A, B, C, D) (*reinterpret_cast<f32x4*>(P) = (f32x4){(A), (B), (C), (D)})
constexpr int kF3Lds = (3 * 64 * kFPS) / 2 + (kAttrCols + 2) * 64 + 64;      // next_w planes [3][64][kFPS] bf16 | WaT rows (+ Wa[:, C], + ba) | bounds

template <bool TRAIN>
__global__ __launch_bounds__(256, 4) void front_fwd3_kernel(FrontFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nprep = g.nprep;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bool late = false;
  if ((int)blockIdx.x < nprep) {
    const int b = blockIdx.x;
    prep_heads_role(g.prep, b % kPrepGridX, (b / kPrepGridX) % kPrepGridY, b / (kPrepGridX * kPrepGridY), lds);
    if (!g.prep_walks) return;
    __syncthreads();
    late = true;
  }
  short* Wp = reinterpret_cast<short*>(lds);                          // next_w as three bf16 planes, row n = output feature
  float* WaT = lds + (3 * 64 * kFPS) / 2;                             // row a < na - 1: attribute_nn.weight[:, a]; na - 1: zeros (padding / foreign ids); na: weight[:, C]; na + 1: bias
  int* bnd = reinterpret_cast<int*>(WaT + (kAttrCols + 2) * 64);
  const int c16 = lane & 15, kq = lane >> 4;
  const int na = g.n_attr;
  {
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = srow + 16 * i;
      const float4 w = *reinterpret_cast<const float4*>(g.Wn + n * 64 + sc4);
      const P3 p0 = split2(w.x, w.y), p1 = split2(w.z, w.w);
      short* d = Wp + n * kFPS + sc4;
      *reinterpret_cast<u32x2*>(d) = (u32x2){p0.h, p1.h}; *reinterpret_cast<u32x2*>(d + 64 * kFPS) = (u32x2){p0.m, p1.m};
      *reinterpret_cast<u32x2*>(d + 2 * 64 * kFPS) = (u32x2){p0.l, p1.l};
    }
    for (int i = tid; i < (na + 2) * 64; i += 256) {
      const int a = i >> 6, c = i & 63;
      WaT[i] = a < na - 1 ? g.Wa[c * na + a] : (a == na - 1 ? 0.f : (a == na ? g.Wa[c * na + na - 1] : g.ba[c]));
    }
    if (tid < na) bnd[tid] = g.attr.bounds[tid];
  }
  __syncthreads();
  const int T = g.count[0];
  const int64_t t_last = (int64_t)T - 1;
  const int ngroups = (T + 15) / 16;
  // group list: rounds over the wavefronts that are walking -- the role blocks join from round g.prep_walks - 1 on (they start late by about
  // what the weight forms take)
  const int W = (int)gridDim.x * 4, Wn_ = W - nprep * 4;               // all wavefronts / the ones without a role
  const int r0 = g.prep_walks > 0 ? g.prep_walks - 1 : 0x3FFFFF;       // first round of the role blocks (never, when they do not walk: small batches)
  const int wv = late ? (int)blockIdx.x * 4 + wave : ((int)blockIdx.x - nprep) * 4 + wave;      // role waves: index among all, others: among the role-free
  const float scale = g.attr.scale;
  const int* ids32 = reinterpret_cast<const int*>(g.ids);
  const float4 bn4 = *reinterpret_cast<const float4*>(g.bn + 4 * kq);          // (+ 16 nb below)
#define F3_GROUP(R) ((R) < r0 ? (late ? 0x3FFFFFF : (R) * Wn_ + wv) : r0 * Wn_ + ((R) - r0) * W + (late ? wv : nprep * 4 + wv))
#define F3_LOAD(GRP, ID, E)                                                                              \
  do {                                                                                                   \
    const int64_t t__ = (int64_t)(GRP) * 16 + c16;                                                       \
    const int64_t tc__ = t__ < t_last ? t__ : t_last;                                                    \
    ID = ids32[2 * tc__];                                                                                \
    const float* src__ = (g.table ? g.table + (int64_t)ID * 64 : g.dense + tc__ * 64) + 8 * kq;          \
    E[0] = *reinterpret_cast<const float4*>(src__); E[1] = *reinterpret_cast<const float4*>(src__ + 4);   \
    E[2] = *reinterpret_cast<const float4*>(src__ + 32); E[3] = *reinterpret_cast<const float4*>(src__ + 36); \
  } while (0)
  int idn = 0;
  float4 en[4];
  const int r_first = late ? r0 : 0;                   // (a role block's first round)
  int grp = F3_GROUP(r_first);
  if (grp < ngroups) F3_LOAD(grp, idn, en);
  for (int r = r_first; grp < ngroups; ++r) {
    const int id = idn;
    float4 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = en[i];
    const int gnext = F3_GROUP(r + 1);
    if (gnext < ngroups) F3_LOAD(gnext, idn, en);        // the next trip's id and row: in flight during this trip
    const int64_t t = (int64_t)grp * 16 + c16;
    // (row of WaT, coordinate) of this lane's token: a search over the <= 31 bounds in LDS, one correctly rounded division (attr_src.hpp)
    const int nb_ = na - 1, top = bnd[nb_];
    const int idc = id < 1 ? 1 : (id > top ? top : id);
    const int ch = attr_chrom(bnd, nb_, idc);
    const bool real = id >= 1 && id <= top;
    const int wrow = real ? ch : nb_;
    const float co = real ? __fdiv_rn((float)(idc - bnd[ch] - 1), scale) : 0.f;
    Frag3 xf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int k0 = 32 * s2 + 8 * kq;
      float v[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float4 wc_ = *reinterpret_cast<const float4*>(&WaT[wrow * 64 + k0 + 4 * hf]);
        const float4 wl_ = *reinterpret_cast<const float4*>(&WaT[na * 64 + k0 + 4 * hf]);
        const float4 ba_ = *reinterpret_cast<const float4*>(&WaT[(na + 1) * 64 + k0 + 4 * hf]);
        const float4 ee = e[2 * s2 + hf];
        v[4 * hf + 0] = (__builtin_fmaf(co, wl_.x, wc_.x) + ba_.x) + ee.x; v[4 * hf + 1] = (__builtin_fmaf(co, wl_.y, wc_.y) + ba_.y) + ee.y;
        v[4 * hf + 2] = (__builtin_fmaf(co, wl_.z, wc_.z) + ba_.z) + ee.z; v[4 * hf + 3] = (__builtin_fmaf(co, wl_.w, wc_.w) + ba_.w) + ee.w;
        if (TRAIN && t < T) FF2_STORE4(g.x0 + t * 64 + k0 + 4 * hf, v[4 * hf], v[4 * hf + 1], v[4 * hf + 2], v[4 * hf + 3]);
      }
      xf[s2] = split8(v);
    }
    // X^T = tanh(Wn x0^T + bn): lane (c16, kq) ends with token c16 and features 16 nb + 4 kq + {0..3}
    // (not unrolled: with the four blocks' A fragments -- 96 registers -- hoisted in front of the MFMAs the kernel spilled; one block's 24 at a time)
#pragma unroll 1
    for (int nb = 0; nb < 4; ++nb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const short* ap = Wp + (16 * nb + c16) * kFPS + 8 * kq;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        Frag3 a;
        a.h = *reinterpret_cast<const u32x4*>(ap + 32 * s2); a.m = *reinterpret_cast<const u32x4*>(ap + 32 * s2 + 64 * kFPS);
        a.l = *reinterpret_cast<const u32x4*>(ap + 32 * s2 + 2 * 64 * kFPS);
        acc = mma6(acc, a, xf[s2]);
      }
      const float4 bb = *reinterpret_cast<const float4*>(g.bn + 16 * nb + 4 * kq);
      if (t < T) FF2_STORE4(g.X + t * 64 + 16 * nb + 4 * kq, fast_tanh(acc[0] + bb.x), fast_tanh(acc[1] + bb.y), fast_tanh(acc[2] + bb.z), fast_tanh(acc[3] + bb.w));
    }
    grp = gnext;
  }
#undef F3_GROUP
#undef F3_LOAD
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
  // attr_mode 1 (the table has get_attributes' structure): front_fwd3_kernel -- no attribute rows gathered, no attribute product; any other
  // table: front_fwd_kernel gathers its rows (padded to one 128-byte unit when the caller padded them) for the K = 32 product
  const bool computed = f.attr_mode == 1 && f.attr_bounds && n_attr <= kAttrCols;
  g.ids = ids; g.table = table; g.dense = dense; g.attr = computed ? attr_src(f, n_attr) : attr_src_table_first(f, n_attr); g.n_attr = n_attr;
  g.Wa = p.attr_w; g.ba = p.attr_b; g.Wn = p.next_w; g.bn = p.next_b; g.count = rg.count; g.x0 = x0; g.X = X;
  const int slots = front_grid() / 2 * 3;               // front_fwd_kernel: three workgroups per CU (44 KB of LDS each)
  const int64_t max_tiles = cdiv(tcap, 64);
  const size_t lds = ((size_t)2 * kTile + 64 * kLdA) * sizeof(float);
  static_assert(((size_t)2 * kTile + 64 * kLdA) >= (size_t)kPrepLdsFloats && kF3Lds >= kPrepLdsFloats, "the weight-form role needs its LDS inside the front end's");
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
    // wave-independent variant: four workgroups per CU, 16-token groups; the role blocks join from round prep_walks - 1 on
    const int slots4 = front_grid() * 2;
    const int64_t ngroups = cdiv(tcap, 16);
    const int64_t rounds = cdiv(ngroups, (int64_t)slots4 * 4);
    g.prep_walks = (g.nprep > 0 && rounds >= 4) ? 3 : 0;                       // (3: the role blocks skip the first two rounds)
    int64_t grid3 = g.prep_walks ? slots4 : g.nprep + (cdiv(ngroups, 4) < slots4 ? cdiv(ngroups, 4) : slots4);
    if (grid3 < g.nprep + 1) grid3 = g.nprep + 1;
    const size_t lds3 = (size_t)kF3Lds * sizeof(float);
    static_assert(kF3Lds >= kPrepLdsFloats, "the weight-form role needs its LDS inside the front end's");
    if (x0) hipLaunchKernelGGL(front_fwd3_kernel<true>, dim3((unsigned)grid3), dim3(256), lds3, st, g);
    else hipLaunchKernelGGL(front_fwd3_kernel<false>, dim3((unsigned)grid3), dim3(256), lds3, st, g);
    MATCHA_CHECK_LAUNCH("front_fwd3_kernel");
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
