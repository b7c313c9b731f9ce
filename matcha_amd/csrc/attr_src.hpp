// The attribute path's input (Modules.py:243-249, :263-264): row `id` of the frozen attribute table [N+1, n_attr].
//
// main.py:497-512 (get_attributes) builds that table as  one-hot(chromosome of the bin) || (index of the bin inside its
// chromosome) / n_0, row 0 = zeros.  When the table handed to the model HAS this structure (the host checks every row, bit for bit:
// matcha_amd/Modules.py::_attr_structure), the kernels do not gather it at all: a token's row is rebuilt from its node id with a
// search over the <= 63 chromosome bounds and one correctly rounded division (`attr_mode` 1) -- one random row per token instead of two
// (SURVEY.md K6).  Any other table is read as rows of `attr_ld` floats (`attr_mode` 0; the host pads the rows to 32 floats = one
// 128-byte fetch unit when n_attr <= 32, so that a row never straddles two).
#pragma once
#include "common.hpp"

namespace matcha {

struct AttrSrc {
  const float* table;        // mode 0: [N+1][ld]
  const int32_t* bounds;     // mode 1: [n_attr] = 0, n_0, n_0 + n_1, ..., N (device); chromosome c owns ids bounds[c]+1 .. bounds[c+1]
  int ld, n_attr, mode;
  float scale;               // mode 1: coordinate = float(id - bounds[c] - 1) / scale
};

static inline AttrSrc attr_src(const matcha_frozen& f, int n_attr) {
  AttrSrc a;
  a.table = f.attr_table; a.bounds = f.attr_bounds; a.n_attr = n_attr; a.mode = f.attr_mode;
  a.ld = f.attr_ld > 0 ? f.attr_ld : n_attr; a.scale = f.attr_scale;
  return a;
}
// for the kernels that keep gathering rows when a table is there (front_fused.hip: its row pieces are loaded by eight threads per row,
// a per-thread rebuild would repeat the search eight times; measured slower than the padded 128-byte rows it replaces)
static inline AttrSrc attr_src_table_first(const matcha_frozen& f, int n_attr) {
  AttrSrc a = attr_src(f, n_attr);
  if (f.attr_table) a.mode = 0;
  return a;
}
static inline int check_attr(const matcha_frozen& f, int n_attr) {
  if (f.attr_mode == 0) {
    MATCHA_CHECK_ARG(f.attr_table, "attribute table missing (attr_mode 0)");
    MATCHA_CHECK_ARG(f.attr_ld == 0 || f.attr_ld >= n_attr, "attr_ld=%d smaller than n_attr=%d", f.attr_ld, n_attr);
  } else {
    MATCHA_CHECK_ARG(f.attr_mode == 1, "attr_mode=%d must be 0 (table rows) or 1 (one-hot chromosome || coordinate, computed)", f.attr_mode);
    MATCHA_CHECK_ARG(f.attr_bounds && n_attr >= 2 && n_attr <= 64 && f.attr_scale > 0.f, "attr_mode 1 needs attr_bounds, 2 <= n_attr <= 64 and attr_scale > 0");
  }
  return MATCHA_OK;
}

// chromosome of node id (1 <= id <= b[nb]): branch-free binary search, the trip count depends on nb only
__device__ __forceinline__ int attr_chrom(const int32_t* __restrict__ b, int nb, int id) {
  int lo = 0, n = nb;
  while (n > 1) {
    const int half = n >> 1;
    lo = (id > b[lo + half]) ? lo + half : lo;
    n -= half;
  }
  return lo;
}

// (chromosome column, coordinate) of a node id under attr_mode 1; id 0 (padding) -> (-1, 0): an all-zero row
__device__ __forceinline__ void attr_decode(const AttrSrc& a, const int32_t* __restrict__ b, int id, int& col, float& coord) {
  const int nb = a.n_attr - 1;
  const int idc = id < 1 ? 1 : (id > b[nb] ? b[nb] : id);
  const int c = attr_chrom(b, nb, idc);
  const bool real = id >= 1 && id <= b[nb];
  col = real ? c : -1;
  coord = real ? __fdiv_rn((float)(idc - b[c] - 1), a.scale) : 0.f;
}

// element `q` of the attribute row described by (col, coord)
__device__ __forceinline__ float attr_elem(int q, int col, float coord, int n_attr) {
  return q == col ? 1.f : ((q == n_attr - 1 && col >= 0) ? coord : 0.f);
}

}  // namespace matcha
