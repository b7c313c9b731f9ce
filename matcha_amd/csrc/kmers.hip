// k-mer generation on the device (reference Code/generate_kmers.py:8-69, one k-mer size per call).
//
// The reference enumerates, for every node i and every cluster containing i, the (k-1)-combinations of the cluster's nodes
// beyond i + min_dis, filters consecutive gaps <= min_dis, counts tuples in a python Counter and keeps those seen at least
// min_freq_cutoff times.  Clusters are sorted unique node lists (process.py:66-77), so that is: every ascending k-subset of
// a cluster whose adjacent gaps all exceed min_dis, counted over clusters, thresholded.  Here:
//   1. one thread per k-subset (combinatorial number system: subset q of cluster c is un-ranked from the binomial table),
//      packed into one integer key of k * bits(n_nodes) bits (64- or 128-bit); subsets failing the gap rule get the all-ones
//      sentinel;
//   2. rocPRIM radix sort of the keys, run-length encode -> (unique k-mer, frequency);
//   3. threshold + un-pack into int64 rows, kept in sorted (lexicographic) order by an exclusive scan of the keep flags.
// The reference's row order depends on worker scheduling; rows here are sorted, which also makes the output reproducible.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kernels.hpp"

namespace matcha {
namespace {

constexpr int kMaxClusterLen = 64;
constexpr int kMaxK = 8;

__global__ void binom_init_kernel(unsigned long long* __restrict__ C) {       // C[n][j], n <= 64, j <= 8
  const int n = threadIdx.x;
  if (n > kMaxClusterLen) return;
  unsigned long long v = 1;
  for (int j = 0; j <= kMaxK; ++j) {
    C[n * (kMaxK + 1) + j] = (j <= n) ? v : 0ull;
    if (j < n) v = v * (unsigned long long)(n - j) / (unsigned long long)(j + 1);
  }
}

template <typename Key>
__global__ __launch_bounds__(256) void kmer_enum_kernel(const int32_t* __restrict__ ids, const int64_t* __restrict__ offsets,
                                                        const int64_t* __restrict__ comb_off, int64_t n_clusters, int64_t total, int k, int bits,
                                                        int min_dis, const unsigned long long* __restrict__ C, Key* __restrict__ keys) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= total) return;
  int64_t lo = 0, hi = n_clusters;                     // last cluster with comb_off[c] <= q
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (comb_off[mid] <= q) lo = mid; else hi = mid;
  }
  const int64_t base = offsets[lo];
  const int n = (int)(offsets[lo + 1] - base);
  unsigned long long rank = (unsigned long long)(q - comb_off[lo]);
  // un-rank the `rank`-th k-subset of {0..n-1} in lexicographic order
  Key key = 0;
  int prev = -1, prev_node = 0;
  bool ok = true;
  for (int p = 0; p < k; ++p) {
    int e = prev + 1;
    for (;; ++e) {
      const unsigned long long c = C[(n - e - 1) * (kMaxK + 1) + (k - p - 1)];     // subsets that start with element e here
      if (rank < c) break;
      rank -= c;
    }
    const int node = ids[base + e];
    if (p > 0 && node - prev_node <= min_dis) ok = false;                           // generate_kmers.py:17, :24-32
    key = (key << bits) | (Key)(unsigned)node;
    prev = e;
    prev_node = node;
  }
  keys[q] = ok ? key : ~(Key)0;
}

template <typename Key>
__global__ __launch_bounds__(256) void kmer_flag_kernel(const Key* __restrict__ uniq, const unsigned int* __restrict__ counts,
                                                        const unsigned int* __restrict__ n_runs, int min_freq, unsigned int* __restrict__ flags,
                                                        int64_t cap_runs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cap_runs) return;
  flags[i] = (i < (int64_t)*n_runs && uniq[i] != ~(Key)0 && counts[i] >= (unsigned)min_freq) ? 1u : 0u;
}

template <typename Key>
__global__ __launch_bounds__(256) void kmer_emit_kernel(const Key* __restrict__ uniq, const unsigned int* __restrict__ counts,
                                                        const unsigned int* __restrict__ flags, const unsigned int* __restrict__ pos,
                                                        const unsigned int* __restrict__ n_runs, int k, int bits, int64_t cap,
                                                        int64_t* __restrict__ out_kmers, int64_t* __restrict__ out_freq, int64_t* __restrict__ n_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t runs = (int64_t)*n_runs;
  if (i == 0) *n_out = runs > 0 ? (int64_t)pos[runs - 1] + flags[runs - 1] : 0;     // total kept (may exceed cap: the caller checks)
  if (i >= runs || !flags[i]) return;
  const int64_t o = pos[i];
  if (o >= cap) return;
  Key key = uniq[i];
  const Key mask = (((Key)1) << bits) - 1;
  for (int p = k - 1; p >= 0; --p) {
    out_kmers[o * k + p] = (int64_t)(unsigned long long)(key & mask);
    key >>= bits;
  }
  out_freq[o] = (int64_t)counts[i];
}

inline int node_bits(int n_nodes) {
  int b = 1;
  while (((int64_t)1 << b) <= (int64_t)n_nodes) ++b;
  return b;
}

template <typename Key>
struct Plan {
  size_t off_binom, off_keys_a, off_keys_b, off_uniq, off_counts, off_runs, off_flags, off_pos, off_tmp, tmp_bytes, total_bytes;
};

template <typename Key>
int make_plan(int64_t total, int end_bit, Plan<Key>& pl) {
  size_t t_sort = 0, t_rle = 0, t_scan = 0;
  Key* kn = nullptr;
  unsigned int* un = nullptr;
  if (rocprim::radix_sort_keys(nullptr, t_sort, kn, kn, (size_t)total, 0, end_bit, (hipStream_t)0) != hipSuccess) return MATCHA_EHIP;
  if (rocprim::run_length_encode(nullptr, t_rle, kn, (unsigned int)total, kn, un, un, (hipStream_t)0) != hipSuccess) return MATCHA_EHIP;
  if (rocprim::exclusive_scan(nullptr, t_scan, un, un, 0u, (size_t)total, rocprim::plus<unsigned int>(), (hipStream_t)0) != hipSuccess) return MATCHA_EHIP;
  size_t tmp = t_sort > t_rle ? t_sort : t_rle;
  if (t_scan > tmp) tmp = t_scan;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  pl.off_binom = take((kMaxClusterLen + 1) * (kMaxK + 1) * sizeof(unsigned long long));
  pl.off_keys_a = take((size_t)total * sizeof(Key));
  pl.off_keys_b = take((size_t)total * sizeof(Key));
  pl.off_uniq = take((size_t)total * sizeof(Key));
  pl.off_counts = take((size_t)total * sizeof(unsigned int));
  pl.off_runs = take(256);
  pl.off_flags = take((size_t)total * sizeof(unsigned int));
  pl.off_pos = take((size_t)total * sizeof(unsigned int));
  pl.off_tmp = take(tmp);
  pl.tmp_bytes = tmp;
  pl.total_bytes = off;
  return MATCHA_OK;
}

template <typename Key>
int run(const int32_t* ids, const int64_t* offsets, const int64_t* comb_off, int64_t n_clusters, int64_t total, int k, int bits, int min_dis,
        int min_freq, int64_t* out_kmers, int64_t* out_freq, int64_t cap, int64_t* n_out, char* ws, size_t ws_bytes, hipStream_t st) {
  Plan<Key> pl;
  const int end_bit = k * bits < (int)(8 * sizeof(Key)) ? k * bits + 1 : (int)(8 * sizeof(Key));   // +1: the sentinel's top bit
  MATCHA_TRY(make_plan<Key>(total, end_bit, pl));
  MATCHA_CHECK_ARG(ws_bytes >= pl.total_bytes, "matcha_kmer_generate: workspace %zu < %zu bytes", ws_bytes, pl.total_bytes);
  unsigned long long* C = reinterpret_cast<unsigned long long*>(ws + pl.off_binom);
  Key* keys_a = reinterpret_cast<Key*>(ws + pl.off_keys_a);
  Key* keys_b = reinterpret_cast<Key*>(ws + pl.off_keys_b);
  Key* uniq = reinterpret_cast<Key*>(ws + pl.off_uniq);
  unsigned int* counts = reinterpret_cast<unsigned int*>(ws + pl.off_counts);
  unsigned int* n_runs = reinterpret_cast<unsigned int*>(ws + pl.off_runs);
  unsigned int* flags = reinterpret_cast<unsigned int*>(ws + pl.off_flags);
  unsigned int* pos = reinterpret_cast<unsigned int*>(ws + pl.off_pos);
  void* tmp = ws + pl.off_tmp;
  size_t tb = pl.tmp_bytes;
  hipLaunchKernelGGL(binom_init_kernel, dim3(1), dim3(128), 0, st, C);
  MATCHA_CHECK_LAUNCH("binom_init_kernel");
  const unsigned blocks = (unsigned)cdiv(total, 256);
  hipLaunchKernelGGL((kmer_enum_kernel<Key>), dim3(blocks), dim3(256), 0, st, ids, offsets, comb_off, n_clusters, total, k, bits, min_dis, C, keys_a);
  MATCHA_CHECK_LAUNCH("kmer_enum_kernel");
  if (rocprim::radix_sort_keys(tmp, tb, keys_a, keys_b, (size_t)total, 0, end_bit, st) != hipSuccess) { set_error("kmer: radix sort failed"); return MATCHA_EHIP; }
  tb = pl.tmp_bytes;
  if (rocprim::run_length_encode(tmp, tb, keys_b, (unsigned int)total, uniq, counts, n_runs, st) != hipSuccess) { set_error("kmer: run-length encode failed"); return MATCHA_EHIP; }
  hipLaunchKernelGGL((kmer_flag_kernel<Key>), dim3(blocks), dim3(256), 0, st, uniq, counts, n_runs, min_freq, flags, total);
  MATCHA_CHECK_LAUNCH("kmer_flag_kernel");
  tb = pl.tmp_bytes;
  if (rocprim::exclusive_scan(tmp, tb, flags, pos, 0u, (size_t)total, rocprim::plus<unsigned int>(), st) != hipSuccess) { set_error("kmer: scan failed"); return MATCHA_EHIP; }
  hipLaunchKernelGGL((kmer_emit_kernel<Key>), dim3(blocks), dim3(256), 0, st, uniq, counts, flags, pos, n_runs, k, bits, cap, out_kmers, out_freq, n_out);
  MATCHA_CHECK_LAUNCH("kmer_emit_kernel");
  return MATCHA_OK;
}

int check(int64_t total, int32_t k, int32_t n_nodes) {
  MATCHA_CHECK_ARG(k >= 2 && k <= kMaxK, "k-mer size %d outside 2..%d", k, kMaxK);
  MATCHA_CHECK_ARG(n_nodes >= 1, "n_nodes must be positive");
  MATCHA_CHECK_ARG(total >= 1 && total < ((int64_t)1 << 32) - 1, "number of candidate k-subsets %lld must be in [1, 2^32 - 2]: split the clusters",
                   (long long)total);
  MATCHA_CHECK_ARG(k * node_bits(n_nodes) <= 127, "k * bits(n_nodes) = %d exceeds the 128-bit key", k * node_bits(n_nodes));
  return MATCHA_OK;
}

}  // namespace
}  // namespace matcha

using namespace matcha;

extern "C" size_t matcha_kmer_workspace_bytes(int64_t total_combos, int32_t k, int32_t n_nodes) {
  if (check(total_combos, k, n_nodes) != MATCHA_OK) return 0;
  const int bits = node_bits(n_nodes);
  if (k * bits <= 63) {
    Plan<unsigned long long> pl;
    if (make_plan<unsigned long long>(total_combos, k * bits + 1, pl) != MATCHA_OK) return 0;
    return pl.total_bytes;
  }
  Plan<__uint128_t> pl;
  if (make_plan<__uint128_t>(total_combos, k * bits + 1, pl) != MATCHA_OK) return 0;
  return pl.total_bytes;
}

extern "C" int matcha_kmer_generate(const int32_t* ids, const int64_t* offsets, const int64_t* comb_off, int64_t n_clusters, int64_t total_combos,
                                    int32_t k, int32_t n_nodes, int32_t min_dis, int32_t min_freq, int64_t* out_kmers, int64_t* out_freq,
                                    int64_t cap, int64_t* n_out, void* ws, size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(ids && offsets && comb_off && out_kmers && out_freq && n_out && ws, "matcha_kmer_generate: null pointer");
  MATCHA_TRY(check(total_combos, k, n_nodes));
  MATCHA_CHECK_ARG(n_clusters >= 1 && cap >= 0 && min_dis >= 0, "matcha_kmer_generate: bad sizes");
  const int bits = node_bits(n_nodes);
  hipStream_t st = (hipStream_t)stream;
  if (k * bits <= 63)
    return run<unsigned long long>(ids, offsets, comb_off, n_clusters, total_combos, k, bits, min_dis, min_freq, out_kmers, out_freq, cap, n_out,
                                   (char*)ws, ws_bytes, st);
  return run<__uint128_t>(ids, offsets, comb_off, n_clusters, total_combos, k, bits, min_dis, min_freq, out_kmers, out_freq, cap, n_out, (char*)ws,
                          ws_bytes, st);
}
