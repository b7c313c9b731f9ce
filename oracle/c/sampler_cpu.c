/* CPU twin (plain C) of matcha_neg_sample (include/matcha_hip.h; matcha_amd/csrc/sampler.hip) -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).
 * It restates, statement for statement, the accept / reject rules of the reference's generate_negative (main.py:361-459) on this
 * project's counter RNG (oracle/rng.py, stream 16), i.e. the same function as oracle/sampler.py -- in C, so that the kernel can be held to it
 * BIT FOR BIT at the bench's batch sizes (the python restatement takes minutes there):
 *   for positive j, negative i (n = neg_num j + i):  the row is resampled only if the positive is a member of the known set (main.py:390-392:
 *   `while neighbor_check(temp, dict)` with temp == the positive on entry; an empty set is the reference's phase 1, main.py:589);
 *   positions: rand(key, n, 0xFFFF0000 + a) & (2^k - 1), redrawn while zero  (Binomial(k, 1/2) != 0 positions, main.py:371-372, :389);
 *   trial t: every chosen position p is replaced by start_c + floor(u (end_c - start_c)), u = rand(key, n, 8 t + p) / 2^32, c = node2chrom of the
 *   ORIGINAL node (main.py:399-407); sort (:416); reject duplicates (:410-414), adjacent gaps <= min_dis (:417-421), known hyperedges (:392).
 * Signature = matcha_neg_sample's minus the stream; `set` (the device hash set) is ignored: membership is decided against `set_edges`
 * through a host hash table built here (exact, like the device's).  status[0] |= 2 for a node without a chromosome, status[1] counts rows
 * whose 65 536 trials were exhausted (returned equal to the positive). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_L 8
#define MAX_TRIALS (1 << 16)
#define STREAM_NEG 16u

static uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
static uint32_t rng_key(uint64_t seed, uint32_t stream) {
  uint32_t k = lowbias32(stream + 0x9E3779B9u);
  k = lowbias32((uint32_t)(seed >> 32) ^ k);
  k = lowbias32((uint32_t)(seed & 0xFFFFFFFFu) ^ k);
  return k;
}
static uint32_t rng_u32(uint32_t key, uint32_t hi, uint32_t lo) { return lowbias32(lo ^ lowbias32(hi ^ key)); }

static uint64_t row_hash(const int64_t* row, int L) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < L; ++i) {
    const uint64_t v = (uint64_t)row[i];
    if (v == 0) break;
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
  }
  return h;
}
static int rows_equal(const int64_t* a, int La, const int64_t* b, int Lb) {
  const int L = La > Lb ? La : Lb;
  for (int i = 0; i < L; ++i) {
    const int64_t va = i < La ? a[i] : 0, vb = i < Lb ? b[i] : 0;
    if (va != vb) return 0;
    if (va == 0) return 1;
  }
  return 1;
}
typedef struct { int64_t cap; int32_t* slots; const int64_t* edges; int L; } HostSet;
static int set_contains(const HostSet* s, const int64_t* row, int L) {
  if (s->cap == 0) return 0;
  uint64_t pos = row_hash(row, L) & (uint64_t)(s->cap - 1);
  for (int64_t probe = 0; probe < s->cap; ++probe) {
    const int32_t idx = s->slots[pos];
    if (idx < 0) return 0;
    if (rows_equal(s->edges + (int64_t)idx * s->L, s->L, row, L)) return 1;
    pos = (pos + 1) & (uint64_t)(s->cap - 1);
  }
  return 0;
}
static int cmp_i64(const void* a, const void* b) {
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

int matcha_neg_sample_cpu(const void* set, const int64_t* set_edges, int64_t n_set_edges, int32_t L_set, const int64_t* pos, int64_t P, int32_t L,
                          int32_t neg_num, int32_t min_dis, const int32_t* node2chrom, int32_t n_nodes, const int32_t* chrom_range,
                          int32_t n_chrom, const uint64_t* seed, int64_t* neg, int32_t* status) {
  (void)set;
  if (!pos || !neg || !node2chrom || !chrom_range || !seed || L < 1 || L > MAX_L || neg_num < 1) return -22;
  HostSet hs = {0, NULL, set_edges, L_set > 0 ? L_set : L};
  if (n_set_edges > 0) {
    hs.cap = 1024;
    while (hs.cap < 2 * n_set_edges) hs.cap <<= 1;
    hs.slots = (int32_t*)malloc((size_t)hs.cap * sizeof(int32_t));
    if (!hs.slots) return -12;
    memset(hs.slots, 0xFF, (size_t)hs.cap * sizeof(int32_t));
    for (int64_t e = 0; e < n_set_edges; ++e) {
      const int64_t* row = set_edges + e * hs.L;
      uint64_t p = row_hash(row, hs.L) & (uint64_t)(hs.cap - 1);
      for (;;) {
        const int32_t idx = hs.slots[p];
        if (idx < 0) { hs.slots[p] = (int32_t)e; break; }
        if (rows_equal(set_edges + (int64_t)idx * hs.L, hs.L, row, hs.L)) break;
        p = (p + 1) & (uint64_t)(hs.cap - 1);
      }
    }
  }
  const uint32_t key = rng_key(*seed, STREAM_NEG);
  for (int64_t n = 0; n < P * neg_num; ++n) {
    const int64_t j = n / neg_num;
    int64_t orig[MAX_L] = {0}, cand[MAX_L] = {0};
    int k = 0;
    for (int i = 0; i < L; ++i) { orig[i] = pos[j * L + i]; if (orig[i] != 0) k = i + 1; }
    int done = 0;
    if (n_set_edges > 0 && k > 0 && set_contains(&hs, orig, L)) {
      uint32_t mask = 0;
      for (uint32_t a = 0; mask == 0; ++a) mask = rng_u32(key, (uint32_t)n, 0xFFFF0000u + a) & ((1u << k) - 1u);
      int64_t cstart[MAX_L], clen[MAX_L];
      for (int i = 0; i < MAX_L; ++i) {
        cstart[i] = 0; clen[i] = -1;
        if (i < k && ((mask >> i) & 1u)) {
          const int in = orig[i] >= 1 && orig[i] <= n_nodes;
          const int c = in ? node2chrom[orig[i]] : -1;
          if (c >= 0 && c < n_chrom) { cstart[i] = chrom_range[2 * c]; clen[i] = (int64_t)chrom_range[2 * c + 1] - cstart[i]; }
          else if (status) status[0] |= 2;            /* kept unchanged; the reference raises KeyError / IndexError (main.py:401-403) */
        }
      }
      for (int trial = 0; trial < MAX_TRIALS && !done; ++trial) {
        for (int i = 0; i < k; ++i) {
          cand[i] = orig[i];
          if (clen[i] >= 0) {
            const uint32_t r = rng_u32(key, (uint32_t)n, (uint32_t)(8 * trial + i));
            cand[i] = cstart[i] + (int64_t)(((uint64_t)r * (uint64_t)clen[i]) >> 32);
          }
        }
        qsort(cand, (size_t)k, sizeof(int64_t), cmp_i64);
        int ok = 1;
        for (int a = 0; a + 1 < k; ++a) {
          const int64_t gap = cand[a + 1] - cand[a];
          if (gap == 0 || gap <= min_dis) ok = 0;
        }
        for (int i = k; i < MAX_L; ++i) cand[i] = 0;
        if (ok && !set_contains(&hs, cand, L)) done = 1;
      }
      if (!done && status) status[1] += 1;
    }
    for (int i = 0; i < L; ++i) neg[n * L + i] = done ? cand[i] : orig[i];
  }
  free(hs.slots);
  return 0;
}
