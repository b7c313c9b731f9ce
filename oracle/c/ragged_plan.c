/* CPU restatement (plain C) of the ragged execution plan of matcha_amd/csrc/ragged.hip -- TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py: nothing under matcha_amd/ links or calls this).
 *
 * What it restates.  The reference pads every hyperedge of a batch to the batch width L with id 0 (pad_sequence, main.py:435-437)
 * and computes every slot; the HIP path compacts the real slots (x != 0, get_non_pad_mask Modules.py:12-14) into a CSR token
 * list and packs whole hyperedges greedily into tiles of at most 63 tokens.  Everything here is integer / index work, so the
 * kernels' output must equal this BIT FOR BIT (tests/test_hip_kernels.py::test_ragged_plan_bit_exact_vs_c_oracle); the same
 * file is compiled with -fsanitize=address,undefined and fuzzed on the host (tests/host/host_checks.hip).
 *
 * Outputs (caller-allocated; T = B*L):
 *   row_off [B+1]  exclusive prefix sum of k_b; row_off[B] = Tr
 *   tok_slot/tok_id/tok_key/tok_pos [T+1]  per compact token: original slot b*L+l, node id (ids outside [0, n_nodes] -> 0 and
 *                  *status |= 1), the id as int32 (0 in unused entries), position-in-hyperedge | k << 8;  entry Tr = the shared
 *                  padding token {B*L, 0, 0, 0}
 *   count [3]      {Tr + 1, Tr, number of tiles}
 *   tile_meta [tiles_cap][4]  {first token, tokens, first hyperedge, hyperedges}; zeros past the tile count
 * Returns the number of tiles, or -1 if tiles_cap is too small. */
#include <stdint.h>
#include <string.h>

#define TILE_TOK 63
#define SUPER_TOK (63 * 32)

int64_t matcha_oracle_ragged_plan(const int64_t* x, int64_t B, int32_t L, int64_t n_nodes, int32_t* row_off, int32_t* tok_slot,
                                  int64_t* tok_id, int32_t* tok_key, int32_t* tok_pos, int32_t* count, int32_t* tile_meta,
                                  int64_t tiles_cap, int32_t* status) {
  const int64_t T = B * (int64_t)L;
  int64_t pos = 0;
  memset(tok_key, 0, (size_t)(T + 1) * sizeof(int32_t));
  for (int64_t b = 0; b < B; ++b) {
    row_off[b] = (int32_t)pos;
    int k = 0;
    for (int l = 0; l < L; ++l) k += x[b * L + l] != 0;
    int nth = 0;
    for (int l = 0; l < L; ++l) {
      int64_t id = x[b * L + l];
      if (id == 0) continue;
      if (id < 0 || id > n_nodes) { if (status) *status |= 1; id = 0; }
      tok_slot[pos] = (int32_t)(b * L + l);
      tok_id[pos] = id;
      tok_key[pos] = (int32_t)id;
      tok_pos[pos] = nth | (k << 8);
      ++pos; ++nth;
    }
  }
  const int64_t Tr = pos;
  row_off[B] = (int32_t)Tr;
  tok_slot[Tr] = (int32_t)T; tok_id[Tr] = 0; tok_pos[Tr] = 0;
  count[0] = (int32_t)(Tr + 1); count[1] = (int32_t)Tr;
  /* tiles: greedy inside each planning superblock = the hyperedges whose FIRST token lies in a window of SUPER_TOK tokens
   * (a hyperedge opens a superblock when its first token lies in another window than its predecessor's first token) */
  memset(tile_meta, 0, (size_t)tiles_cap * 4 * sizeof(int32_t));
  int64_t nt = 0;
  int64_t b = 0;
  while (b < B) {
    /* this superblock: [b, b_hi) -- up to the next hyperedge whose start window differs from its predecessor's */
    int64_t b_hi = b + 1;
    while (b_hi < B && row_off[b_hi] / SUPER_TOK == row_off[b_hi - 1] / SUPER_TOK) ++b_hi;
    int64_t tile_b0 = b;
    int32_t tile_tok0 = row_off[b];
    for (int64_t i = b; i < b_hi; ++i) {
      while (row_off[i + 1] - tile_tok0 > TILE_TOK) {      /* hyperedge i ends beyond the open tile: it opens the next one */
        if (nt >= tiles_cap) return -1;
        tile_meta[4 * nt + 0] = tile_tok0; tile_meta[4 * nt + 1] = row_off[i] - tile_tok0;
        tile_meta[4 * nt + 2] = (int32_t)tile_b0; tile_meta[4 * nt + 3] = (int32_t)(i - tile_b0);
        ++nt;
        tile_b0 = i; tile_tok0 = row_off[i];
      }
    }
    if (nt >= tiles_cap) return -1;
    tile_meta[4 * nt + 0] = tile_tok0; tile_meta[4 * nt + 1] = row_off[b_hi] - tile_tok0;
    tile_meta[4 * nt + 2] = (int32_t)tile_b0; tile_meta[4 * nt + 3] = (int32_t)(b_hi - tile_b0);
    ++nt;
    b = b_hi;
  }
  count[2] = (int32_t)nt;
  return nt;
}

/* The half tiles and the token -> (tile, row) map of the wave-independent fused forward (matcha_amd/csrc/fused_fwd32.hip):
 *   half_meta [halves_cap][4]  the same greedy packing inside the same superblocks with at most HALF_TOK = 31 tokens per half tile
 *                              (+ the shared padding token = 32 rows, one wavefront); zeros past the count
 *   tok_tile [T+1]             (tile << 6) | row for every token of tile_meta's tiles (entries of no tile are left untouched)
 * row_off / tile_meta / n_tiles are the outputs of matcha_oracle_ragged_plan.  Returns the number of half tiles (= count[3]),
 * or -1 if halves_cap is too small. */
#define HALF_TOK 31

int64_t matcha_oracle_ragged_halves(const int32_t* row_off, int64_t B, const int32_t* tile_meta, int64_t n_tiles, int32_t* half_meta,
                                    int64_t halves_cap, int32_t* tok_tile) {
  memset(half_meta, 0, (size_t)halves_cap * 4 * sizeof(int32_t));
  int64_t nh = 0;
  int64_t b = 0;
  while (b < B) {
    int64_t b_hi = b + 1;
    while (b_hi < B && row_off[b_hi] / SUPER_TOK == row_off[b_hi - 1] / SUPER_TOK) ++b_hi;
    int64_t tile_b0 = b;
    int32_t tile_tok0 = row_off[b];
    for (int64_t i = b; i < b_hi; ++i) {
      while (row_off[i + 1] - tile_tok0 > HALF_TOK) {
        if (nh >= halves_cap) return -1;
        half_meta[4 * nh + 0] = tile_tok0; half_meta[4 * nh + 1] = row_off[i] - tile_tok0;
        half_meta[4 * nh + 2] = (int32_t)tile_b0; half_meta[4 * nh + 3] = (int32_t)(i - tile_b0);
        ++nh;
        tile_b0 = i; tile_tok0 = row_off[i];
      }
    }
    if (nh >= halves_cap) return -1;
    half_meta[4 * nh + 0] = tile_tok0; half_meta[4 * nh + 1] = row_off[b_hi] - tile_tok0;
    half_meta[4 * nh + 2] = (int32_t)tile_b0; half_meta[4 * nh + 3] = (int32_t)(b_hi - tile_b0);
    ++nh;
    b = b_hi;
  }
  for (int64_t w = 0; w < n_tiles; ++w)
    for (int32_t i = 0; i < tile_meta[4 * w + 1]; ++i) tok_tile[tile_meta[4 * w] + i] = (int32_t)((w << 6) | i);
  return nh;
}
