// K16: negative sampling (main.py:361-459) on the GPU against an EXACT hash set of the known hyperedges.
//
// The reference keeps one Bloom filter per hyperedge size (utils.py:75-97, pybloom_live) and rejects a
// candidate when `tuple(temp) in filter`.  Here membership is exact: an open-addressing table of int32
// indices into the (immutable) edge list; a probe compares the candidate with the stored row itself.
// A zero-padded ascending row [a,b,c,0,...] identifies (k, tuple) uniquely because node ids are >= 1.
//
// Randomness is the counter RNG of oracle/rng.py, stream STREAM_NEG; negative n = neg_num*j + i of positive j
// draws   mask bits   : rand(key, n, 0xFFFF0000 + attempt)  -> low k bits, redrawn while zero
//                       (each position chosen with prob 1/2, conditioned on >= 1 chosen  ==  the reference's
//                        Binomial(k, 1/2) != 0 count (main.py:371-372) + uniform choice of positions (:389))
//         replacement : rand(key, n, 8*trial + position)     -> start + floor(u * (end - start))    (:405-407)
#include "kernels.hpp"

namespace matcha {

// Set layout (round 6): 64 int32 header words (words 0..1 = capacity, a power of two), then `capacity` slots of 8 bytes: {int32 index of the
// stored hyperedge in the edge list (-1 = empty), uint32 tag = upper 32 bits of the row's hash}.  A probe reads a WINDOW of eight consecutive
// slots in one round trip (linear probing: the probe chain of a row lies inside its window with probability 1 - load^8) and compares tags;
// only a tag match costs a second trip (the stored row itself: exactness does not rest on the tag).  Rounds 1-5 kept bare indices and loaded a
// stored row per occupied slot: slot -> row -> next slot -> row ..., two dependent trips per probe, and the slowest lane of a wavefront paid
// a chain of ~5 probes -- the negative sampler was a chain of 10+ global round trips (18.8 us for 288 negatives).
constexpr int kSetHeader = 64;         // int32 words reserved in front of the slots (word 0..1 = capacity)
constexpr int kSetWindow = 8;          // slots per probe window
constexpr int kMaxTrials = 1 << 16;
struct SetSlot { int32_t idx; uint32_t tag; };
static_assert(sizeof(SetSlot) == 8, "a slot is one 8-byte word (inserted with a 64-bit compare-and-swap)");
__device__ __forceinline__ uint32_t row_tag(uint64_t h) { return (uint32_t)(h >> 32); }

__device__ __forceinline__ uint64_t row_hash(const int64_t* __restrict__ row, int L) {
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
// rows equal as zero-padded tuples of possibly different widths
__device__ __forceinline__ bool rows_equal(const int64_t* __restrict__ a, int La, const int64_t* __restrict__ b, int Lb) {
  const int L = La > Lb ? La : Lb;
  for (int i = 0; i < L; ++i) {
    const int64_t va = i < La ? a[i] : 0, vb = i < Lb ? b[i] : 0;
    if (va != vb) return false;
    if (va == 0) return true;
  }
  return true;
}

// The same two functions on a row held in REGISTERS (neg_sample_kernel's candidate): every index is a compile-time constant after
// unrolling, so the array never becomes a scratch-memory object (dynamic indices cost 144 B of scratch per lane and 94 scratch
// instructions in the first version); the early exits are flags.
__device__ __forceinline__ uint64_t row_hash_reg(const int64_t (&row)[MATCHA_MAX_L], int L) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  bool live = true;
#pragma unroll
  for (int i = 0; i < MATCHA_MAX_L; ++i) {
    const uint64_t v = (uint64_t)row[i];
    live = live && i < L && v != 0;
    uint64_t t = h ^ (v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    t *= 0xBF58476D1CE4E5B9ull;
    t ^= t >> 31;
    h = live ? t : h;
  }
  return h;
}
__device__ __forceinline__ bool rows_equal_reg(const int64_t* __restrict__ a, int La, const int64_t (&b)[MATCHA_MAX_L], int Lb) {
  const int L = La > Lb ? La : Lb;
  bool decided = false, equal = true;
#pragma unroll
  for (int i = 0; i < MATCHA_MAX_L; ++i) {
    const int64_t va = (i < La) ? a[i < La ? i : 0] : 0, vb = i < Lb ? b[i] : 0;
    const bool on = !decided && i < L;
    if (on && va != vb) { decided = true; equal = false; }
    if (on && va == vb && va == 0) decided = true;
  }
  return equal;
}
__device__ __forceinline__ bool set_contains_reg(const int32_t* __restrict__ set, const int64_t* __restrict__ edges, int L_set,
                                                 const int64_t (&row)[MATCHA_MAX_L], int L) {
  const int64_t cap = reinterpret_cast<const int64_t*>(set)[0];
  const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(set + kSetHeader);
  const uint64_t h = row_hash_reg(row, L);
  const uint32_t tag = row_tag(h);
  uint64_t pos = h & (uint64_t)(cap - 1);
  for (int64_t probe = 0; probe < cap; probe += kSetWindow) {
    unsigned long long w[kSetWindow];
#pragma unroll
    for (int i = 0; i < kSetWindow; ++i) w[i] = slots[(pos + (uint64_t)i) & (uint64_t)(cap - 1)];      // eight loads in flight: one round trip
    bool open = true;                                    // no empty slot met yet
#pragma unroll
    for (int i = 0; i < kSetWindow; ++i) {
      const int32_t idx = (int32_t)(uint32_t)(w[i] & 0xFFFFFFFFull);
      const uint32_t t = (uint32_t)(w[i] >> 32);
      if (open && idx < 0) open = false;
      if (open && t == tag && rows_equal_reg(edges + (int64_t)idx * L_set, L_set, row, L)) return true;
    }
    if (!open) return false;
    pos = (pos + kSetWindow) & (uint64_t)(cap - 1);
  }
  return false;
}

__global__ void hashset_clear_kernel(int32_t* set, int64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { reinterpret_cast<int64_t*>(set)[0] = cap; }
  if (i < cap) reinterpret_cast<unsigned long long*>(set + kSetHeader)[i] = ~0ull;        // idx = -1, tag = all ones
}

__global__ void hashset_insert_kernel(int32_t* set, const int64_t* __restrict__ edges, int64_t n, int L) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t cap = reinterpret_cast<const int64_t*>(set)[0];
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(set + kSetHeader);
  const int64_t* row = edges + i * L;
  const uint64_t h = row_hash(row, L);
  const unsigned long long mine = ((unsigned long long)row_tag(h) << 32) | (unsigned long long)(uint32_t)(int32_t)i;
  uint64_t pos = h & (uint64_t)(cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    const unsigned long long old = atomicCAS(&slots[pos], ~0ull, mine);
    if (old == ~0ull) return;                                      // inserted
    const int32_t oidx = (int32_t)(uint32_t)(old & 0xFFFFFFFFull);
    if ((uint32_t)(old >> 32) == row_tag(h) && rows_equal(edges + (int64_t)oidx * L, L, row, L)) return;   // duplicate row already present
    pos = (pos + 1) & (uint64_t)(cap - 1);
  }
}

__device__ __forceinline__ bool set_contains(const int32_t* __restrict__ set, const int64_t* __restrict__ edges, int L_set,
                                             const int64_t* row, int L) {
  const int64_t cap = reinterpret_cast<const int64_t*>(set)[0];
  const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(set + kSetHeader);
  const uint64_t h = row_hash(row, L);
  const uint32_t tag = row_tag(h);
  uint64_t pos = h & (uint64_t)(cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    const unsigned long long w = slots[pos];
    const int32_t idx = (int32_t)(uint32_t)(w & 0xFFFFFFFFull);
    if (idx < 0) return false;
    if ((uint32_t)(w >> 32) == tag && rows_equal(edges + (int64_t)idx * L_set, L_set, row, L)) return true;
    pos = (pos + 1) & (uint64_t)(cap - 1);
  }
  return false;
}

__global__ void hashset_contains_kernel(const int32_t* __restrict__ set, const int64_t* __restrict__ edges, int L_set,
                                        const int64_t* __restrict__ rows, int64_t n, int L, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = set_contains(set, edges, L_set, rows + i * L, L) ? 1 : 0;
}

// one thread per negative
__global__ __launch_bounds__(256) void neg_sample_kernel(const int32_t* __restrict__ set, const int64_t* __restrict__ set_edges,
                                                         int64_t n_set, int L_set, const int64_t* __restrict__ pos, int64_t P,
                                                         int L, int neg_num, int min_dis, const int32_t* __restrict__ node2chrom, int n_nodes,
                                                         const int32_t* __restrict__ chrom_range, int n_chrom,
                                                         const uint64_t* __restrict__ seed, int64_t* __restrict__ neg, int32_t* __restrict__ status) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P * neg_num) return;
  const int64_t j = n / neg_num;
  int64_t orig[MATCHA_MAX_L], cand[MATCHA_MAX_L];
  int k = 0;
#pragma unroll
  for (int i = 0; i < MATCHA_MAX_L; ++i) {
    orig[i] = (i < L) ? pos[j * L + i] : 0;
    if (orig[i] != 0) k = i + 1;
  }
  int64_t* out = neg + n * L;
  // The chromosome range of EVERY position is looked up before the membership test of the positive, not after it and only for the
  // positions drawn: node -> chromosome -> range is a chain of two dependent loads that then runs next to the hash probe's two
  // instead of behind them (this kernel is a chain of global-memory round trips; at the reference's 288 negatives per step it is
  // nothing else).  Clamped indices, no branch around the loads.
  const uint64_t seed_v = *seed;
  int cidx[MATCHA_MAX_L];
#pragma unroll
  for (int i = 0; i < MATCHA_MAX_L; ++i) {
    const bool in = i < k && orig[i] >= 1 && orig[i] <= n_nodes;
    const int c = node2chrom[in ? orig[i] : 0];
    cidx[i] = (in && c >= 0 && c < n_chrom) ? c : -1;
  }
  int2 crange[MATCHA_MAX_L];
#pragma unroll
  for (int i = 0; i < MATCHA_MAX_L; ++i) crange[i] = reinterpret_cast<const int2*>(chrom_range)[cidx[i] >= 0 ? cidx[i] : 0];
  // `while neighbor_check(temp, dict)` with temp == the positive on entry (main.py:390-392): if the positive is
  // not a member (in particular: empty set, the reference's phase 1, main.py:589) the loop never runs.
  bool resample = (n_set > 0) && (k > 0) && set_contains_reg(set, set_edges, L_set, orig, L);
  bool done = false;
  if (resample) {
    const uint32_t key = rng_key(seed_v, kStreamNeg);
    uint32_t mask = 0;
    for (uint32_t a = 0; mask == 0; ++a) mask = rng_u32(key, (uint32_t)n, 0xFFFF0000u + a) & ((1u << k) - 1u);
    // the chromosome ranges of the positions that may be redrawn do not change between trials
    int64_t cstart[MATCHA_MAX_L], clen[MATCHA_MAX_L];
#pragma unroll
    for (int i = 0; i < MATCHA_MAX_L; ++i) {
      cstart[i] = 0; clen[i] = -1;
      if (i < k && ((mask >> i) & 1u)) {
        // a node outside [1, n_nodes] or without a chromosome (node2chrom = -1, the default fill of train.run) cannot be
        // redrawn: it is kept and the call is flagged (the reference raises KeyError / IndexError at main.py:401-403)
        if (cidx[i] >= 0) {
          cstart[i] = crange[i].x;
          clen[i] = (int64_t)crange[i].y - cstart[i];
        } else if (status) {
          atomicOr(status, MATCHA_STATUS_BAD_CHROM);
        }
      }
    }
    for (int trial = 0; trial < kMaxTrials && !done; ++trial) {
      // unused slots (i >= k) sort behind every node id and are put back to 0 afterwards
      constexpr int64_t kBig = 0x7FFFFFFFFFFFFFFFll;
#pragma unroll
      for (int i = 0; i < MATCHA_MAX_L; ++i) {
        cand[i] = i < k ? orig[i] : kBig;
        if (clen[i] >= 0) {
          const uint32_t r = rng_u32(key, (uint32_t)n, (uint32_t)(8 * trial + i));
          cand[i] = cstart[i] + (int64_t)(((uint64_t)r * (uint64_t)clen[i]) >> 32);
        }
      }
      // sort ascending (k <= 8): Batcher's odd-even merge network, 19 compare-exchanges on registers (an insertion sort indexes the array
      // with run-time values); then reject duplicates / close neighbours / known hyperedges (main.py:410-421, :392)
#define NS_CE(A, B) do { const int64_t lo__ = cand[A] < cand[B] ? cand[A] : cand[B], hi__ = cand[A] < cand[B] ? cand[B] : cand[A]; cand[A] = lo__; cand[B] = hi__; } while (0)
      static_assert(MATCHA_MAX_L == 8, "the sorting network below is for 8 slots");
      NS_CE(0, 1); NS_CE(2, 3); NS_CE(4, 5); NS_CE(6, 7);
      NS_CE(0, 2); NS_CE(1, 3); NS_CE(4, 6); NS_CE(5, 7);
      NS_CE(1, 2); NS_CE(5, 6);
      NS_CE(0, 4); NS_CE(1, 5); NS_CE(2, 6); NS_CE(3, 7);
      NS_CE(2, 4); NS_CE(3, 5);
      NS_CE(1, 2); NS_CE(3, 4); NS_CE(5, 6);
#undef NS_CE
      bool ok = true;
#pragma unroll
      for (int a = 0; a + 1 < MATCHA_MAX_L; ++a) {
        const int64_t gap = cand[a + 1] - cand[a];
        if (a + 1 < k && (gap == 0 || gap <= min_dis)) ok = false;
      }
#pragma unroll
      for (int i = 0; i < MATCHA_MAX_L; ++i) cand[i] = i < k ? cand[i] : 0;
      if (ok && !set_contains_reg(set, set_edges, L_set, cand, L)) done = true;
    }
    // trials exhausted (tiny chromosome, large min_dis, dense known set): the row is returned equal to its positive and counted;
    // the reference would loop forever (main.py:392)
    if (!done && status) atomicAdd(status + 1, 1);
  }
#pragma unroll
  for (int i = 0; i < MATCHA_MAX_L; ++i)
    if (i < L) out[i] = done ? cand[i] : orig[i];
}

}  // namespace matcha

using namespace matcha;

static int64_t set_capacity(int64_t n_edges) {
  int64_t cap = 1024;
  while (cap < 2 * n_edges) cap <<= 1;
  return cap;
}

extern "C" size_t matcha_hashset_bytes(int64_t n_edges) {
  return (size_t)kSetHeader * sizeof(int32_t) + (size_t)set_capacity(n_edges < 0 ? 0 : n_edges) * sizeof(SetSlot);
}

extern "C" int matcha_hashset_build(void* set, size_t set_bytes, const int64_t* edges, int64_t n_edges, int32_t L,
                                    matcha_stream_t stream) {
  MATCHA_CHECK_ARG(set && (edges || n_edges == 0), "matcha_hashset_build: null pointer");
  MATCHA_CHECK_ARG(L >= 1 && L <= MATCHA_MAX_L, "matcha_hashset_build: L=%d", L);
  MATCHA_CHECK_ARG(n_edges >= 0 && n_edges < (1ll << 31), "matcha_hashset_build: n_edges=%lld", (long long)n_edges);
  MATCHA_CHECK_ARG(set_bytes >= matcha_hashset_bytes(n_edges), "matcha_hashset_build: set buffer too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t cap = set_capacity(n_edges);
  hipLaunchKernelGGL(hashset_clear_kernel, dim3((unsigned)cdiv(cap, 256)), dim3(256), 0, st, (int32_t*)set, cap);
  MATCHA_CHECK_LAUNCH("hashset_clear_kernel");
  if (n_edges > 0) {
    hipLaunchKernelGGL(hashset_insert_kernel, dim3((unsigned)cdiv(n_edges, 256)), dim3(256), 0, st, (int32_t*)set, edges, n_edges, L);
    MATCHA_CHECK_LAUNCH("hashset_insert_kernel");
  }
  return MATCHA_OK;
}

extern "C" int matcha_hashset_contains(const void* set, const int64_t* edges, int32_t L_set, const int64_t* rows, int64_t n,
                                       int32_t L, int32_t* out, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(set && rows && out, "matcha_hashset_contains: null pointer");
  MATCHA_CHECK_ARG(L >= 1 && L <= MATCHA_MAX_L && L_set >= 1 && L_set <= MATCHA_MAX_L, "matcha_hashset_contains: bad width");
  if (n <= 0) return MATCHA_OK;
  hipLaunchKernelGGL(hashset_contains_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const int32_t*)set,
                     edges, L_set, rows, n, L, out);
  MATCHA_CHECK_LAUNCH("hashset_contains_kernel");
  return MATCHA_OK;
}

extern "C" int matcha_neg_sample(const void* set, const int64_t* set_edges, int64_t n_set_edges, int32_t L_set,
                                 const int64_t* pos, int64_t P, int32_t L, int32_t neg_num, int32_t min_dis,
                                 const int32_t* node2chrom, int32_t n_nodes, const int32_t* chrom_range, int32_t n_chrom,
                                 const uint64_t* seed, int64_t* neg, int32_t* status, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(pos && neg && node2chrom && chrom_range && seed, "matcha_neg_sample: null pointer");
  MATCHA_CHECK_ARG(((uintptr_t)chrom_range) % 8 == 0, "matcha_neg_sample: chrom_range must be 8-byte aligned (the kernel reads [lo, hi] pairs as one int2)");
  MATCHA_CHECK_ARG(n_nodes >= 1 && n_chrom >= 1, "matcha_neg_sample: n_nodes=%d n_chrom=%d", n_nodes, n_chrom);
  MATCHA_CHECK_ARG(n_set_edges == 0 || (set && set_edges), "matcha_neg_sample: non-empty set without buffers");
  MATCHA_CHECK_ARG(L >= 1 && L <= MATCHA_MAX_L && neg_num >= 1, "matcha_neg_sample: L=%d neg_num=%d", L, neg_num);
  if (P <= 0) return MATCHA_OK;
  ProfScope ps(MATCHA_PROF_NEG_SAMPLE, (double)P * L * 8.0 * (1.0 + neg_num), (hipStream_t)stream);
  hipLaunchKernelGGL(neg_sample_kernel, dim3((unsigned)cdiv(P * neg_num, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const int32_t*)set, set_edges, n_set_edges, L_set > 0 ? L_set : L, pos, P, L, neg_num, min_dis, node2chrom, n_nodes,
                     chrom_range, n_chrom, seed, neg, status);
  MATCHA_CHECK_LAUNCH("neg_sample_kernel");
  return MATCHA_OK;
}
