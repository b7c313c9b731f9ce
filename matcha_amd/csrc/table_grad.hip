// Deterministic embedding backward (nn.Embedding's weight gradient, Modules.py:29-34; SURVEY.md §7 "Scatter-add: sort/segment
// by node id first") and the reduce side of the row-sparse data-parallel exchange.
//
// Input: a list of n (node id, gradient row) pairs -- the compact token list of one backward pass, or the all-gathered lists
// of every rank back to back.  The pairs are stably sorted by id (LSD radix sort over the ceil(log2(N + 2)) id bits), and one
// 16-lane group per RUN of equal ids adds the run's rows in list order and adds the sum to dtable[id] with plain stores: one
// writer per table row, a fixed order of additions -> the table gradient is bitwise reproducible, with no float atomics
// (which are ~75-way contended per row at hg38 1 Mb sizes and execute at the memory side at <= 1.3 TB/s, MI355X_MICROARCH.md).
// Entries with id 0 (padding / unused list slots) sort behind every real id and are skipped.
//
// The sort is rocPRIM's device radix sort (AMD's own primitive library, headers only); the segmented sum is hand-written.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "kernels.hpp"

namespace matcha {

namespace {

__host__ __device__ inline int id_bits(int n_nodes) {
  int b = 1;
  while ((1ll << b) < (long long)n_nodes + 2) ++b;     // the sentinel 2^b - 1 must exceed every id
  return b;
}

// sort key of list entry i: its id, or the all-ones sentinel for id 0 / ids outside [1, n_nodes]
__global__ __launch_bounds__(256) void tg_keys_kernel(const int32_t* __restrict__ ids, int64_t n, int n_nodes, uint32_t sentinel,
                                                      uint32_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t id = ids[i];
  keys[i] = (id >= 1 && id <= n_nodes) ? (uint32_t)id : sentinel;
}

// One 16-lane group per sorted position; the group of a run's FIRST position walks the run (4 rows in flight) and adds its
// rows in sorted (= list) order.  NCH float4 chunks per lane: d = 64 * NCH at most.
template <int NCH>
__global__ __launch_bounds__(256) void tg_segsum_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ src, int64_t n, int d,
                                                        uint32_t sentinel, const float* __restrict__ rows, float* __restrict__ dtable) {
  const int s = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (key == sentinel || (i > 0 && keys[i - 1] == key)) return;       // not the head of a run of a real id
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t j = i;
  for (;;) {
    // up to 4 list entries of this run per trip: their row loads are independent, the adds keep list order
    uint32_t kk[4], ss[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t q = j + u < n ? j + u : n - 1;
      kk[u] = (j + u < n) ? keys[q] : sentinel;
      ss[u] = src[q];
    }
    float4 v[4][NCH];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = 4 * s + 64 * c;
        v[u][c] = (kk[u] == key && col < d) ? *reinterpret_cast<const float4*>(rows + (int64_t)ss[u] * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    bool more = true;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (kk[u] != key) { more = false; break; }
#pragma unroll
      for (int c = 0; c < NCH; ++c) { acc[c].x += v[u][c].x; acc[c].y += v[u][c].y; acc[c].z += v[u][c].z; acc[c].w += v[u][c].w; }
    }
    if (!more) break;
    j += 4;
  }
  float* dst = dtable + (int64_t)key * d;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = 4 * s + 64 * c;
    if (col < d) {
      float4 o = *reinterpret_cast<float4*>(dst + col);
      o.x += acc[c].x; o.y += acc[c].y; o.z += acc[c].z; o.w += acc[c].w;
      *reinterpret_cast<float4*>(dst + col) = o;
    }
  }
}

struct TgWs {
  uint32_t *keys_in, *keys_out, *src;
  void* tmp;
  size_t tmp_bytes, total;
};

size_t tg_carve(int64_t n, int n_nodes, char* base, TgWs& w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  w.keys_in = (uint32_t*)take((size_t)n * 4);
  w.keys_out = (uint32_t*)take((size_t)n * 4);
  w.src = (uint32_t*)take((size_t)n * 4);
  size_t tb = 0;
  rocprim::counting_iterator<uint32_t> iota(0);
  const hipError_t rc = rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, iota, (uint32_t*)nullptr, (size_t)n, 0u,
                                                  (unsigned)id_bits(n_nodes), (hipStream_t)0);
  // rocPRIM picks its configuration from the current device; on a machine without one (the CPU build container, where only
  // the sizing queries run) the query fails: reserve a bound no configuration exceeds (double buffers of keys and values +
  // histograms) so that the workspace layout stays well defined
  if (rc != hipSuccess || tb == 0) { (void)hipGetLastError(); tb = (size_t)n * 16 + (1u << 20); }
  w.tmp_bytes = tb;
  w.tmp = take(tb);
  w.total = off;
  return off;
}

}  // namespace

size_t table_grad_ws_bytes(int64_t n, int n_nodes) {
  if (n <= 0) return 256;
  TgWs w;
  return tg_carve(n, n_nodes, nullptr, w);
}

int launch_table_grad(const int32_t* ids, const float* rows, int64_t n, int d, int n_nodes, float* dtable, void* ws, size_t ws_bytes,
                      hipStream_t st) {
  if (n <= 0) return MATCHA_OK;
  TgWs w;
  const size_t need = tg_carve(n, n_nodes, (char*)ws, w);
  if (ws_bytes < need) { set_error("table gradient: workspace %zu < %zu bytes", ws_bytes, need); return MATCHA_ENOMEM; }
  const int bits = id_bits(n_nodes);
  const uint32_t sentinel = (uint32_t)((1ull << bits) - 1ull);
  // read ids + rows once, add 4d bytes per real entry (backward scatter-add = k * 4d bytes added)
  ProfScope ps(MATCHA_PROF_EMBED_SCATTER, (double)n * (4.0 + 8.0 * d), st);
  hipLaunchKernelGGL(tg_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, ids, n, n_nodes, sentinel, w.keys_in);
  MATCHA_CHECK_LAUNCH("tg_keys_kernel");
  rocprim::counting_iterator<uint32_t> iota(0);
  size_t tb = w.tmp_bytes;
  if (rocprim::radix_sort_pairs(w.tmp, tb, w.keys_in, w.keys_out, iota, w.src, (size_t)n, 0u, (unsigned)bits, st) != hipSuccess) {
    set_error("table gradient: radix sort failed");
    return MATCHA_EHIP;
  }
  const dim3 grid((unsigned)cdiv(n, 16));
  if (d <= 64) hipLaunchKernelGGL((tg_segsum_kernel<1>), grid, dim3(256), 0, st, w.keys_out, w.src, n, d, sentinel, rows, dtable);
  else if (d <= 128) hipLaunchKernelGGL((tg_segsum_kernel<2>), grid, dim3(256), 0, st, w.keys_out, w.src, n, d, sentinel, rows, dtable);
  else hipLaunchKernelGGL((tg_segsum_kernel<4>), grid, dim3(256), 0, st, w.keys_out, w.src, n, d, sentinel, rows, dtable);
  MATCHA_CHECK_LAUNCH("tg_segsum_kernel");
  return MATCHA_OK;
}

}  // namespace matcha

using namespace matcha;

extern "C" size_t matcha_scatter_rows_workspace_bytes(int64_t n, int32_t d, int32_t n_nodes) {
  (void)d;
  return table_grad_ws_bytes(n, n_nodes);
}

extern "C" int matcha_scatter_rows(const int32_t* ids, const float* rows, int64_t n, int32_t d, int32_t n_nodes, float* dtable,
                                   void* ws, size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(ids && rows && dtable && ws, "matcha_scatter_rows: null pointer");
  MATCHA_CHECK_ARG(n >= 0 && n < (1ll << 31), "matcha_scatter_rows: n=%lld", (long long)n);
  MATCHA_CHECK_ARG(d >= 4 && d <= 256 && d % 4 == 0, "matcha_scatter_rows: d=%d must be a multiple of 4, <= 256", d);
  MATCHA_CHECK_ARG(n_nodes >= 1 && n_nodes < (1 << 30), "matcha_scatter_rows: n_nodes=%d", n_nodes);
  MATCHA_CHECK_ARG(((uintptr_t)ws) % 256 == 0, "matcha_scatter_rows: workspace must be 256-byte aligned");
  return launch_table_grad(ids, rows, n, d, n_nodes, dtable, ws, ws_bytes, (hipStream_t)stream);
}
