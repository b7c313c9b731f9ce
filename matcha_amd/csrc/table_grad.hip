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
// Two sorters.  Up to 32 767 nodes (every hg38 layout: 3 067 bins at 1 Mb, 30 344 at 100 kb) the id IS the digit of a single
// hand-written counting-sort pass: per tile of 4 096 list entries a histogram (LDS atomics), column / id scans, then up to four
// wavefronts per tile keep one counter per node in LDS, find the lanes that hold the same id with `bits` wave ballots (no
// per-lane loop), and write each entry to  start[id] + (entries of the id in earlier tiles) + (earlier entries of this tile)
// -- a stable sort in 4 launches, after which
// one WAVEFRONT per node adds its run with 16 rows in flight.  rocPRIM's generic radix sort took 12 launches (115 us) for the
// 327 681 entries of the bench batch and the one-group-per-position sum 78 us; this path: see DESIGN.md §4.2.  Larger tables
// (BASELINE config 5: 1 M nodes, runs of length ~1) keep rocPRIM's device radix sort + the one-group-per-position sum.
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

// ---- counting sort with the node id as the digit (n_nodes < 32 768) -----------------------------------------------------------
constexpr int kCsTile = 4096;          // list entries per wavefront
constexpr int kCsMaxNodes = 32767;     // counters of one wavefront: (n_nodes + 1) ints of LDS <= 128 KB

// hist[blk][id] = entries of id in tile blk.  Counting needs no order: LDS integer atomics, 256 threads per tile.
__global__ __launch_bounds__(256) void tg_hist_kernel(const int32_t* __restrict__ ids, int64_t n, int n_nodes, int stride, int32_t* __restrict__ hist) {
  extern __shared__ int cnt[];
  const int64_t blk = blockIdx.x;
  for (int i = threadIdx.x; i < stride; i += 256) cnt[i] = 0;
  __syncthreads();
  const int64_t t0 = blk * kCsTile;
#pragma unroll 4
  for (int it = threadIdx.x; it < kCsTile; it += 256) {
    const int64_t t = t0 + it;
    const int32_t id = t < n ? ids[t] : 0;
    if (id >= 1 && id <= n_nodes) atomicAdd(&cnt[id], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < stride; i += 256) hist[blk * stride + i] = cnt[i];
}

// src[position] = entry index, position = start[id] + base[blk][id] + (entries of the id earlier in the tile): a STABLE
// placement (ranks follow list order).  W wavefronts per tile, wave w owns the w-th part of the tile and its own LDS counters:
//   1. every wave counts its part (LDS atomics on its own counters -- counts are order-free);
//   2. counters become bases: start + base[blk] + the counts of the waves before;
//   3. every wave ranks its part 64 entries at a time: the lanes that hold the same id find each other with `bits` wave
//      ballots, rank = number of lower lanes among them, the lowest one advances the counter.
template <int W>
__global__ __launch_bounds__(64 * W) void tg_place_kernel(const int32_t* __restrict__ ids, int64_t n, int n_nodes, int bits, int stride,
                                                          const int32_t* __restrict__ base, const int32_t* __restrict__ start,
                                                          uint32_t* __restrict__ src) {
  extern __shared__ int lds_cnt[];
  constexpr int kPart = kCsTile / W;                                     // entries per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t blk = blockIdx.x;
  int* cnt = lds_cnt + wave * stride;
  const int64_t t0 = blk * kCsTile + (int64_t)wave * kPart;
  if (W > 1) {
    for (int i = lane; i < stride; i += 64) cnt[i] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): this wave's own zeroes have landed (in-order LDS, one wave)
#pragma unroll 4
    for (int it = lane; it < kPart; it += 64) {
      const int64_t t = t0 + it;
      const int32_t id = t < n ? ids[t] : 0;
      if (id >= 1 && id <= n_nodes) atomicAdd(&cnt[id], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < stride; i += 64 * W) {                 // exclusive prefix over the waves, plus the global bases
      int run = start[i] + base[blk * stride + i];
#pragma unroll
      for (int w = 0; w < W; ++w) { const int c = lds_cnt[w * stride + i]; lds_cnt[w * stride + i] = run; run += c; }
    }
    __syncthreads();
  } else {
    for (int i = lane; i < stride; i += 64) cnt[i] = start[i] + base[blk * stride + i];
    __syncthreads();
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  constexpr int kBatch = 16;                                             // ids of 16 wave-steps are loaded before the first is ranked:
  for (int it0 = 0; it0 < kPart; it0 += 64 * kBatch) {                   // one global-memory round trip per 1 024 entries, not per 64
    int32_t idb[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int64_t t = t0 + it0 + 64 * u + lane;
      idb[u] = t < n ? ids[t] : 0;
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int64_t t = t0 + it0 + 64 * u + lane;
      const int32_t id = idb[u];
      const int key = (id >= 1 && id <= n_nodes) ? id : 0;               // 0: unused entry
      uint64_t peers = __ballot(key != 0);                               // lanes that hold the same id as this one
      for (int b = 0; b < bits; ++b) {
        const bool bit = (key >> b) & 1;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
      }
      if (key != 0) {
        const int rank = __popcll(peers & lt);
        const int old = cnt[key];                                        // every peer reads the counter before the leader moves it
        src[old + rank] = (uint32_t)t;
        if (rank == 0) cnt[key] = old + __popcll(peers);                 // leaders hold distinct ids: no conflict
      }
    }
  }
}

// hist[blk][id] -> exclusive prefix over the tiles (in place), total[id]
__global__ __launch_bounds__(256) void tg_colscan_kernel(int32_t* __restrict__ hist, int nblk, int stride, int32_t* __restrict__ total) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= stride) return;
  int run = 0;
  int b = 0;
  for (; b + 16 <= nblk; b += 16) {                                      // 16 independent loads in flight, then the serial prefix
    int c[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) c[u] = hist[(int64_t)(b + u) * stride + v];
#pragma unroll
    for (int u = 0; u < 16; ++u) { hist[(int64_t)(b + u) * stride + v] = run; run += c[u]; }
  }
  for (; b < nblk; ++b) {
    const int c = hist[(int64_t)b * stride + v];
    hist[(int64_t)b * stride + v] = run;
    run += c;
  }
  total[v] = run;
}

// start[id] = exclusive prefix sum of total[] (one workgroup; stride <= 32 768 = 1024 threads x 32); start[stride] = sum
__global__ __launch_bounds__(1024) void tg_idscan_kernel(const int32_t* __restrict__ total, int stride, int32_t* __restrict__ start) {
  __shared__ int part[1024];
  const int per = (stride + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = (lo + per < stride) ? lo + per : stride;
  int local = 0;
  for (int i = lo; i < hi; ++i) local += total[i];
  part[threadIdx.x] = local;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - local;
  for (int i = lo; i < hi; ++i) { start[i] = run; run += total[i]; }
  if (threadIdx.x == 1023) start[stride] = part[1023];
}

// one wavefront per node: four 16-lane groups take the entries j = g, g + 4, ... of the node's run (two rows in flight each) and
// their partial sums are combined as (g0 + g1) + (g2 + g3): a fixed order for a given list
template <int NCH>
__global__ __launch_bounds__(256) void tg_runsum_kernel(const int32_t* __restrict__ start, const uint32_t* __restrict__ src, int n_nodes, int d,
                                                        const float* __restrict__ rows, float* __restrict__ dtable) {
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;                // node id 1 .. n_nodes
  if (v > n_nodes) return;
  const int lo = start[v], hi = start[v + 1];
  if (lo == hi) return;
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t n0 = lo + g < hi ? src[lo + g] : 0u, n1 = lo + g + 4 < hi ? src[lo + g + 4] : 0u;
  for (int j = lo + g; j < hi; j += 8) {
    const bool two = j + 4 < hi;
    const uint32_t e0 = n0, e1 = two ? n1 : n0;
    n0 = j + 8 < hi ? src[j + 8] : 0u;                                   // next trip's entry indices: off the critical path
    n1 = j + 12 < hi ? src[j + 12] : 0u;
    float4 a[NCH], b[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * s + 64 * c;
      a[c] = col < d ? *reinterpret_cast<const float4*>(rows + (int64_t)e0 * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[c] = (two && col < d) ? *reinterpret_cast<const float4*>(rows + (int64_t)e1 * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      acc[c].x += a[c].x; acc[c].y += a[c].y; acc[c].z += a[c].z; acc[c].w += a[c].w;
      if (two) { acc[c].x += b[c].x; acc[c].y += b[c].y; acc[c].z += b[c].z; acc[c].w += b[c].w; }
    }
  }
  // (g0 + g1) + (g2 + g3): xor 16 pairs g0/g1 and g2/g3, xor 32 pairs the two sums
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float* f = reinterpret_cast<float*>(&acc[c]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float o1 = __shfl_xor(f[q], 16, 64);
      const float s1 = (g & 1) ? o1 + f[q] : f[q] + o1;                 // both lanes of a pair compute lower + upper
      const float o2 = __shfl_xor(s1, 32, 64);
      f[q] = (g & 2) ? o2 + s1 : s1 + o2;
    }
  }
  if (g == 0) {
    float* dst = dtable + (int64_t)v * d;
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
}

struct TgWs {
  uint32_t *keys_in, *keys_out, *src;
  void* tmp;
  size_t tmp_bytes, total;
  // counting-sort path
  int32_t *hist, *totals, *start;
  int nblk, stride;
};

size_t tg_carve(int64_t n, int n_nodes, char* base, TgWs& w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  if (n_nodes <= kCsMaxNodes) {
    w.nblk = (int)cdiv(n, kCsTile);
    w.stride = n_nodes + 1;
    w.src = (uint32_t*)take((size_t)n * 4);
    w.hist = (int32_t*)take((size_t)w.nblk * w.stride * 4);
    w.totals = (int32_t*)take((size_t)w.stride * 4);
    w.start = (int32_t*)take((size_t)(w.stride + 1) * 4);
    w.keys_in = w.keys_out = nullptr; w.tmp = nullptr; w.tmp_bytes = 0;
    w.total = off;
    return off;
  }
  w.hist = w.totals = w.start = nullptr; w.nblk = w.stride = 0;
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
  if (n_nodes <= kCsMaxNodes) {
    const size_t lds = (size_t)w.stride * sizeof(int);
    const int waves = 4 * lds <= 150 * 1024 ? 4 : (2 * lds <= 150 * 1024 ? 2 : 1);     // LDS counters per wavefront: (n_nodes + 1) ints
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tg_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tg_place_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tg_place_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tg_place_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(tg_hist_kernel, dim3(w.nblk), dim3(256), lds, st, ids, n, n_nodes, w.stride, w.hist);
    MATCHA_CHECK_LAUNCH("tg_hist_kernel");
    hipLaunchKernelGGL(tg_colscan_kernel, dim3((unsigned)cdiv(w.stride, 256)), dim3(256), 0, st, w.hist, w.nblk, w.stride, w.totals);
    MATCHA_CHECK_LAUNCH("tg_colscan_kernel");
    hipLaunchKernelGGL(tg_idscan_kernel, dim3(1), dim3(1024), 0, st, w.totals, w.stride, w.start);
    MATCHA_CHECK_LAUNCH("tg_idscan_kernel");
    if (waves == 4) hipLaunchKernelGGL((tg_place_kernel<4>), dim3(w.nblk), dim3(256), 4 * lds, st, ids, n, n_nodes, bits, w.stride, w.hist, w.start, w.src);
    else if (waves == 2) hipLaunchKernelGGL((tg_place_kernel<2>), dim3(w.nblk), dim3(128), 2 * lds, st, ids, n, n_nodes, bits, w.stride, w.hist, w.start, w.src);
    else hipLaunchKernelGGL((tg_place_kernel<1>), dim3(w.nblk), dim3(64), lds, st, ids, n, n_nodes, bits, w.stride, w.hist, w.start, w.src);
    MATCHA_CHECK_LAUNCH("tg_place_kernel");
    const dim3 grid((unsigned)cdiv(n_nodes, 4));
    if (d <= 64) hipLaunchKernelGGL((tg_runsum_kernel<1>), grid, dim3(256), 0, st, w.start, w.src, n_nodes, d, rows, dtable);
    else if (d <= 128) hipLaunchKernelGGL((tg_runsum_kernel<2>), grid, dim3(256), 0, st, w.start, w.src, n_nodes, d, rows, dtable);
    else hipLaunchKernelGGL((tg_runsum_kernel<4>), grid, dim3(256), 0, st, w.start, w.src, n_nodes, d, rows, dtable);
    MATCHA_CHECK_LAUNCH("tg_runsum_kernel");
    return MATCHA_OK;
  }
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
