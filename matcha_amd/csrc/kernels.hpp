// Internal launcher prototypes shared by the translation units of libmatcha_hip.so.
#pragma once
#include "common.hpp"

namespace matcha {

struct GemmArgs {
  const float* A[3];
  const float* B[3];
  float* C[3];
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int batch;
  // epilogue (applied in the order documented in include/matcha_hip.h)
  int flags;
  const float* bias[3];
  const float* residual;
  const float* aux;
  const int64_t* row_ids;
  const uint64_t* seed;
  uint32_t stream_id;
  float p_drop;
  float aux_scale;
  // optional indirections (adj front end; all device pointers, null = off)
  const int32_t* m_dev;       // number of rows lives on the device (rows >= *m_dev are neither read nor written)
  const int32_t* a_row_map;   // A row of logical row m is a_row_map[m]
  const int32_t* c_row_map;   // C (and residual/aux/row-mask) row of logical row m is c_row_map[m]
  const int32_t* seg;         // grouped mode: logical rows are sorted by group; group c owns rows [seg[c], seg[c+1])
  int n_groups;               //   and uses B + c*b_group_stride; rows of groups >= n_groups are skipped
  int64_t b_group_stride;
  const int32_t* rng_row_map; // dropout counter row of C row m is rng_row_map[m] (compact token -> original slot)
};

// ragged.hip: CSR plan of the real tokens of x [B, L] (+ one shared padding token at index Tr)
struct Ragged {
  int32_t* row_off;    // [B+1]
  int32_t* tok_slot;   // [T+1]
  int64_t* tok_id;     // [T+1]
  int32_t* count;      // {Tr + 1, Tr, tiles, half tiles}
  int32_t* blk_sum;
  int nblk;
  int ntiles;          // capacity of tile_meta (upper bound of the tile count; the count itself is count[2])
  int32_t* tok_pos;    // [T+1] position of the token inside its hyperedge | k << 8
  int32_t* tok_key;    // [T+1] node id as int32, 0 in unused slots (table-gradient list)
  int32_t* tile_meta;  // [ntiles + 2][4] {first token t0, number of tokens, first hyperedge b0, number of hyperedges}; zeros past the end
  int32_t* sb_tiles;   // planning scratch: per-superblock tile lists
  int32_t* sb_cnt;
  int32_t* sb_first;   // [nsb + 1] first hyperedge of each planning superblock (B where none starts)
  int nsb, sb_cap;
  int32_t* half_meta;  // [nhalves + 2][4] half tiles (<= 31 tokens of whole hyperedges), same fields as tile_meta; count[3] of them
  int32_t* tok_tile;   // [T+1] (tile << 6) | row of the token inside tile_meta's tiling
  int32_t* sb_htiles;  // planning scratch of the half tiles
  int32_t* sb_hcnt;
  int nhalves, sb_hcap;
};
size_t ragged_bytes(int64_t B, int L);
int ragged_tiles_cap(int64_t B, int L);
int ragged_halves_cap(int64_t B, int L);
void ragged_carve(int64_t B, int L, char* base, Ragged& r);
int launch_ragged_plan(const int64_t* x, int64_t B, int L, int64_t n_nodes, int32_t* status, const Ragged& r, hipStream_t st, int level = 2);

struct HeadParams {
  const float *gp, *bp, *g1, *b1, *g2, *b2, *wc, *bc;
};

// gemm_f32.hip
int launch_gemm_rm(bool b_kn, const GemmArgs& g, hipStream_t st);
size_t gemm_tn_ws_bytes(int64_t M, int64_t N, int64_t R);
int launch_gemm_tn(const float* A, const float* B, float* C, float* colsum, int64_t M, int64_t N, int64_t R, int64_t lda,
                   int64_t ldb, const int64_t* b_gather, bool accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                   const int32_t* r_dev = nullptr, const float* out_scale_dev = nullptr, float out_scale = 1.f);   // C (+)= scale * A^T B
int launch_slab_reduce(const float* slab, float* out1, int64_t n1, float* out2, int64_t stride, int P, bool accumulate, hipStream_t st,
                       const float* scale_dev = nullptr, float scale = 1.f);

// bmm_heads.hip: up to four d x d x d products x 8 heads in one launch, out_h[i][j] (+)= sum_x A_h(i, x) B_h(x, j); an operand element is
// ptr[head * hs + row * rs + col * cs] (one of rs, cs is 1), C is row-major with row stride c_rs
struct BmmProduct {
  const float* A; int64_t a_rs, a_cs, a_hs;
  const float* B; int64_t b_rs, b_cs, b_hs;
  float* C; int64_t c_rs, c_hs;
  int accumulate;
};
bool bmm_heads_supported(int d);
int launch_bmm_heads(const BmmProduct* prods, int n, int d, hipStream_t st);

// gemm_wide.hip: 128 x 128 workgroup tiles for embed_dim >= 128 (returns / eligibility: see the file header)
bool gemm_wide_eligible(bool b_kn, const GemmArgs& g);
int launch_gemm_wide(bool b_kn, const GemmArgs& g, hipStream_t st);
bool gemm_tn_wide_eligible(int64_t M, int64_t N, int64_t R, int64_t lda, int64_t ldb, const float* A, const float* B, const int64_t* b_gather);
size_t gemm_tn_wide_ws_bytes(int64_t M, int64_t N, int64_t R);
int launch_gemm_tn_wide(const float* A, const float* B, int64_t M, int64_t N, int64_t R, int64_t lda, int64_t ldb, bool colsum, void* ws,
                        size_t ws_bytes, const int32_t* r_dev, int* P_out, int64_t* slab_stride_out, hipStream_t st);

// token_kernels.hip
int launch_embed_fwd(const int64_t* x, int64_t T, int d, const float* table, const float* dense, const matcha_frozen& f,
                     int n_attr, const float* Wa, const float* ba, float* x0, hipStream_t st, const int32_t* t_dev = nullptr);
int launch_embed_scatter(const int64_t* x, int64_t T, int d, const float* dx0, float* dtable, hipStream_t st, const int32_t* t_dev = nullptr);
int launch_gather_rows(const int64_t* ids, int64_t T, int d, const float* table, int64_t n_nodes, float* rows, int32_t* status, hipStream_t st);
int launch_check_ids(const int64_t* ids, int64_t T, int64_t n_nodes, int32_t* status, hipStream_t st);
int launch_expand_embedding(const int64_t* x, int64_t B, int L, int d, const int32_t* row_off, const float* H2, const float* X, const float* gp,
                            const float* bp, float* dynamic, float* static_, hipStream_t st);

// table_grad.hip: deterministic embedding backward -- stable sort of (id, row) pairs by id + one writer per table row
size_t table_grad_ws_bytes(int64_t n, int n_nodes);
int launch_table_grad(const int32_t* ids, const float* rows, int64_t n, int d, int n_nodes, float* dtable, void* ws, size_t ws_bytes, hipStream_t st);

// process-wide A/B switches (matcha_set_option; initial values from the environment, read once)
struct Options {
  int disable_fused;         // 1: layer-by-layer kernels everywhere; 2: the front end only (encoder stays fused)
  int disable_merged;        // the reference formulation of the heads (four products per head) on the layer-by-layer kernels: the A/B variant
  int disable_small_batch;   // the large-batch kernels at every size: one wavefront per half tile in the forward, the ragged plan as five launches
  int disable_wide_gemm;     // embed_dim >= 128: the 64-wide GEMM / attention kernels
  int debug_nan, fused_dbg;  // development
};
Options& options();
int launch_fill_i32(int32_t* p, int n, int32_t v, hipStream_t st);
// Zero `bytes` bytes (a multiple of 4, 4-byte aligned) with a KERNEL.  The library does not use hipMemsetAsync on paths that callers capture
// into hipGraphs: a small memset NODE of a captured training step was observed to run unordered with the kernel node that consumes the
// buffer (wrong reconstruction-head gradients in replayed steps only; tools/debug/graph_vs_eager.py), kernel nodes keep stream order.
int zero_async(void* p, size_t bytes, hipStream_t st);
int launch_ln3_fwd(const float* X, int64_t T, int d, const float* gq, const float* bq, const float* gk, const float* bk,
                   const float* gv, const float* bv, float* qin, float* kin, float* vin, float* stats, hipStream_t st,
                   const int32_t* t_dev = nullptr);
int launch_ln3_bwd(const float* X, const float* dqin, const float* dkin, const float* dvin, const float* dXs, int64_t T, int d,
                   const float* gq, const float* gk, const float* gv, float* dZ0, float* slab, float* dgq, float* dbq,
                   float* dgk, float* dbk, float* dgv, float* dbv, hipStream_t st, const int32_t* t_dev = nullptr);
int launch_head_fwd(const int32_t* row_off, const float* H2, const float* X, int64_t B, int L, int d, const HeadParams& hp,
                    const float* y, const float* w, float* logits, float* row_loss, float* bce_out, hipStream_t st);
int launch_head_bwd(const int32_t* row_off, const float* H2, const float* X, int64_t B, int L, int d, const HeadParams& hp,
                    const float* y, const float* w, const float* logits, const float* dlogits, float alpha, float* dH2,
                    float* dXs, float* slab, const HeadParams& ghp, hipStream_t st);
size_t colsum_slab_bytes(int64_t n, int nv, int d);
int launch_loss_reduce(const float* row_loss, int64_t B, float* bce_out, hipStream_t st, bool zero_recon = false,     // zero_recon: losses[1..2] = 0 too
                       float* zero_buf = nullptr, size_t zero_bytes = 0);                                               // a buffer zeroed by extra blocks of the launch

// fused_aux.hip (embed_dim 64): per-step weight folding, reduction of the training forward's parameter-gradient slabs
size_t fused_fold_floats();
size_t fused_tail_slab_floats(int64_t B, int L);
size_t fused_tail_partial_floats();
// two passes over one slab per (half) tile (the forward with its tail's backward in-kernel at a large batch); small batches and the split tail
// are summed by blocks of fbm_reduce_kernel's launch (tail_reduce.hpp, launch_fused_bwd_merged)
int launch_tail_reduce(const float* tslab, const Ragged& rg, int L, matcha_tensors& grads, hipStream_t st, bool halves, float* partial);
bool fused_small_batch(const Ragged& rg);          // the size rule of the small-batch kernels (fused_fwd32h_kernel)

// fused_fwd32.hip (embed_dim 64): the same forward with ONE wavefront per half tile (<= 31 tokens), weights streamed from L2 in
// MFMA-fragment order (launch_prep_heads rewrites them once per step), no workgroup barriers
size_t fused_frag_floats();
size_t fused_tail_slab32_floats(int64_t B, int L);
// the heads in their two-products form (r = B_h x + b_h, dyn += M_h z); `merged` = fused_merged_floats() floats of workspace that
// launch_prep_heads fills (B [8][64][64], M [8][64][64], b [8][64], bdyn [64])
size_t fused_merged_floats();
struct MergedView { const float* B; const float* M; const float* bvec; const float* bdyn; };
MergedView merged_view(const float* merged);
int launch_prep_heads(const matcha_tensors& p, float* folded, float* merged, float* frag, hipStream_t st);     // the three per-step weight forms, one launch
int launch_fused_fwd32(const matcha_tensors& p, const float* folded, const float* frag, const float* X, const Ragged& rg, int64_t B, int L, const float* y,
                       const float* w, float* Y, float* H1, float* H2, float* logits, float* row_loss, const uint64_t* seed, float p_fc1, float p_pff,
                       hipStream_t st, float* ddyn0 = nullptr, float* dXs = nullptr, float* tslab = nullptr, float alpha = 0.f, float* rimg = nullptr,
                       float* tail_dh2 = nullptr);       // tail_dh2 (large batches only): the convolutions' backward is left to launch_tail_bwd64
// tail_bwd.hip: the backward of pff_n1's two convolutions as its own kernel behind fused_fwd32_kernel (large batches)
int tail_bwd_grid();
size_t tail_bwd_slab_floats();
int launch_tail_bwd64(const matcha_tensors& p, const float* dH2, const float* Y, const float* H1, const Ragged& rg, const uint64_t* seed, float p_fc1,
                      float p_pff, float* ddyn0, float* slab, const float* vslab, float* zero_rows, hipStream_t st,
                      const float* row_loss = nullptr, int64_t B = 0, float* losses = nullptr, bool zero_recon = false);   // losses != null: launch_loss_reduce's work in one extra block

// fused_bwd.hip (embed_dim 64): attention-block backward from X and dDyn; accumulates the gradients of w_q/w_k/w_v, the
// three LayerNorm affines in front of them, fc1 (weight + bias) and writes dZ0 (gradient at the next_w pre-activation)
size_t fused_bwd_ws_floats(int64_t B, int L);
struct TailReduceArgs;
int launch_fused_bwd_merged(const matcha_tensors& p, const float* folded, const float* merged, const float* X, const float* dDyn, const float* dXs,
                            const Ragged& rg, int64_t B, int L, float* dxh, float* ws, matcha_tensors& grads, float* dZ0, hipStream_t st, const float* rimg,
                            bool dx_atomic, bool dx_zeroed = false,    // dx_zeroed: the caller already zeroed dxh[(B L + 1) x 64] on this stream
                            const struct TailReduceArgs* tail = nullptr);   // tail != null: the launch that sums this kernel's slabs also sums the forward's tail slabs (tail_reduce.hpp)
size_t fused_qkv_floats(int64_t B, int L);         // what the training forward leaves for the fused backward, per (half tile, head):
constexpr int kImgRecH = 2048 + 256;               // 32 r rows (r = B_h x_hat + b_h; register images) + their attention probabilities [32][8].
                                                   // (Round 6 measured the alternative -- the forward keeps only the probabilities, fused_bwdh_kernel recomputes r from its
                                                   // staged x_hat planes, 24 more MFMAs per half tile: forward -19 us, backward +52 us on one box, same call: reverted.)

const float* fused_bwd_dxpad(const float* ws);     // d x_hat of the shared padding token inside the fused backward's workspace

// enc128.hip (embed_dim 128): the attention block (three LayerNorms, merged heads, attention, fc1 + dropout) as ONE forward and ONE backward
// kernel in x_hat space with the LayerNorm affines folded into the merged matrices; r / z / dR / dZ never reach HBM
bool enc128_shape(int d);
size_t enc128_ws_floats();
size_t enc128_rec_floats(const Ragged& rg);        // what a training forward leaves for the backward: r rows + probabilities per (half tile, head)
int launch_enc128_fwd(const matcha_tensors& p, const float* lwB, const float* lwM, const float* X, const Ragged& rg, int64_t B, int L, float* Y, float* rec,
                      float* ws, const int32_t* tok_slot, const uint64_t* seed, float p_drop, hipStream_t st);
int launch_enc128_bwd(const matcha_tensors& p, const float* lwB, const float* lwM, const float* X, const float* dDyn, const float* dXs, const Ragged& rg,
                      int64_t B, int L, float* dxh, const float* rec, float* ws, float* lwdB, float* lwdM, matcha_tensors& grads, float* dZ0,
                      hipStream_t st);

// front_fused.hip (embed_dim 64, n_attr <= 32): forward = gather + attribute_nn + next_w + tanh in one kernel;
// backward = LayerNorm backward of the summed d x_hat partials, next_w and attribute_nn
// backward, embedding scatter (dtable != null) or dX0 output (adj front end) in one kernel
bool front_bwd_supported(int d, int n_attr);
// prep != null: the launch also builds the encoder's per-step weight forms (what launch_prep_heads does) in blocks of their own
struct PrepSpec { const matcha_tensors* p; float* folded; float* merged; float* frag; };
int launch_front_fwd(const matcha_tensors& p, const int64_t* ids, const float* table, const float* dense, const matcha_frozen& f, int n_attr,
                     const Ragged& rg, int64_t tcap, float* x0, float* X, hipStream_t st, const PrepSpec* prep = nullptr);
size_t front_bwd_ws_floats();
int launch_front_bwd(const matcha_tensors& p, const float* X, const float* dxh, int nslab, int64_t tcap, const float* dxpad, const float* dXs, const float* x0,
                     const int64_t* ids, const matcha_frozen& f, int n_attr, const Ragged& rg, float* dX0, float* dtable, float* ws,
                     matcha_tensors& grads, hipStream_t st, int32_t* touched = nullptr);

// attention.hip
// shared_kv (embed_dim >= 128, merged heads): K and V are [T, d] tensors that every head attends (dK / dV still come back per head)
int launch_attn_fwd(const float* Q, const float* K, const float* V, const int32_t* row_off, int64_t B, int L, int d, float* O, float* P,
                    hipStream_t st, bool shared_kv = false);
int launch_attn_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, const int32_t* row_off, int64_t B, int L,
                    int d, float* dQ, float* dK, float* dV, float* slab, hipStream_t st, bool shared_kv = false, float* dKsum = nullptr,
                    float* dVsum = nullptr);
size_t attn_bwd_slab_bytes(int64_t B, int d);
// attention_wide.hip: the same attention for embed_dim >= 128 with one operand set live at a time (no spills at L = 8, d = 256)
bool attn_wide_eligible(int d);
int launch_attn_fwd_wide(const float* Q, const float* K, const float* V, const int32_t* row_off, int64_t B, int L, int d, float inv_temp, float* O,
                         float* P, int nblk, hipStream_t st, bool shared_kv = false);
int launch_attn_bwd_wide(const float* Q, const float* K, const float* V, const float* P, const float* dO, const int32_t* row_off, int64_t B, int L,
                         int d, float inv_temp, float* dQ, float* dK, float* dV, float* slab, int nblk, hipStream_t st, bool shared_kv = false,
                         float* dKs = nullptr, float* dVs = nullptr);

}  // namespace matcha
