// Helpers shared by the adj front end's kernels (adj_frontend.hip: layer-by-layer; adj_fused.hip: the fused d = 64 path).
#pragma once
#include "kernels.hpp"

namespace matcha {

// Four consecutive floats of a row of n floats that starts at an arbitrary float offset (weight rows of a chromosome: n_c floats each).
// gfx950 serves a global_load_dwordx4 on a 4-byte-aligned address (tools/ubench/unaligned_x4.hip: correct, same cost as four dword loads
// when bandwidth-bound), and one such load per lane keeps 4x the bytes in flight of the scalar staging it replaces.  A window that would
// run past the row's end is read from the row's last four floats and shifted (row4_fix); columns >= n come back as 0.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f4u row4_load(const float* __restrict__ row, int col0, int n) {
  if (n >= 4) {
    const int start = col0 < n - 4 ? col0 : n - 4;
    return *reinterpret_cast<const f4u*>(row + start);
  }
  f4u v;
  v.x = row[0]; v.y = row[n > 1 ? 1 : 0]; v.z = row[n > 2 ? 2 : 0]; v.w = 0.f;       // n in 1..3: the whole row
  return v;
}
__device__ __forceinline__ void row4_fix(const f4u& raw, int col0, int n, float (&e)[4]) {
  e[0] = raw.x; e[1] = raw.y; e[2] = raw.z; e[3] = raw.w;
  int sh = n >= 4 ? (col0 < n - 4 ? 0 : col0 - (n - 4)) : col0;                          // floats the window was moved back by
  if (sh >= 4) { e[0] = e[1] = e[2] = e[3] = 0.f; return; }
  if (sh == 1) { e[0] = e[1]; e[1] = e[2]; e[2] = e[3]; e[3] = 0.f; }
  else if (sh == 2) { e[0] = e[2]; e[1] = e[3]; e[2] = 0.f; e[3] = 0.f; }
  else if (sh == 3) { e[0] = e[3]; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
  if (n < 4) {                                                                           // short rows: zero what lies behind the end
    if (col0 + 0 >= n) e[0] = 0.f;
    if (col0 + 1 >= n) e[1] = 0.f;
    if (col0 + 2 >= n) e[2] = 0.f;
    if (col0 + 3 >= n) e[3] = 0.f;
  }
}

constexpr int kMaxChrom = 63;     // buckets = chromosomes + 1 (padding)

// workspace of the adj front end (adj_frontend.hip::adj_carve)
struct AdjWs {
  int32_t *order, *other_map, *seg, *counts /* [0]=m (other tokens), [1]=non-pad tokens */, *hist, *base;
  float *Hs, *TH, *rec, *dTH, *dZ, *lossslab;
  float* rgrad;        // fused path: UNSCALED gradient of the reconstruction head of this step's chromosome, [nr_pad][d] weight | [nr_pad] bias
  int nblk;
  int64_t nr_pad;
  size_t total;
};

// row stride (floats) of chromosome c's feature matrix: n_c, or n_c rounded up to the padding unit (matcha_frozen.feat_row_pad)
__host__ __device__ __forceinline__ int feat_ld(int n_c, int pad) { return pad > 0 ? (n_c + pad - 1) / pad * pad : n_c; }

// adj_fused.hip: the fused kernels of embed_dim 64
bool adj_fused_eligible(const matcha_shape& s, const matcha_frozen& f);
int adj_fused_forward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o, const int64_t* ids, int64_t T,
                      const AdjWs& w, int r_chrom, bool save, float* node_out, float* x0, float* X, float* recon_out, hipStream_t st,
                      const int32_t* slot_map);
int adj_fused_backward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o, const int64_t* ids, int64_t T,
                       const AdjWs& w, int r_chrom, const float* dX0, const float* drecon, matcha_tensors& g, hipStream_t st,
                       const int32_t* slot_map, int32_t* touched);

}  // namespace matcha
